#!/bin/bash
# Build every native piece from any cwd.
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()"
