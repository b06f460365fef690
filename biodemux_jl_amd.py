"""Import shim: the package directory is named ``biodemux.jl_amd`` (with a dot), which the
import system cannot spell; this module loads it under the importable name
``biodemux_jl_amd`` (sub-modules resolve normally: ``biodemux_jl_amd.hipabi`` ...)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "biodemux.jl_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

if __name__ == "__main__":  # python biodemux_jl_amd.py <fastq1> <barcode_file> <output_directory> [options] (cli.jl)
    from biodemux_jl_amd.cli import main as _main

    _sys.exit(_main())
