"""Summary reports of ``summary=true`` runs — the host-side consumer of the DemuxStats the device collects.

Mirrors BioDemuX.jl src/reporting.jl: ``generate_summary_report`` (:567-577) dispatches on
``config.summary_format`` to the text (:60-130), JSON (:132-232), stdout (:234-299) or HTML (:301-565) writer.
Same files (``summary.txt`` / ``summary.json`` / ``summary.html`` in the output directory), same append
behaviour for repeated runs (a separator line; a JSON list that grows; run sections added before ``</body>``),
same field names and line texts, so the expectations of the reference's test/integration/summary_mode.jl and
summary_distributions.jl hold verbatim.  Numbers are printed the way Julia prints them (shortest round-trip
floats, ``round(x, digits=2)`` percentages, ``Dates.canonicalize`` durations).  The HTML page is this
repository's own layout (tables + inline SVG bar charts), not the reference's template.
"""
from __future__ import annotations

import datetime as _dt
import html as _html
import math
import os
import sys
from typing import Dict, Optional

from .classification import DemuxStats, _round2


# ---- Julia-style printing ----
def jl_float(x: float) -> str:
    """``string(x::Float64)``: shortest round-trip digits; exponent form (1.0e-5, 1.0e6) outside 1e-4 <= |x| < 1e6."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    r = repr(x)
    if "e" not in r and abs(x) >= 1e6:  # Julia switches to the exponent form at 1e6 (Python at 1e16)
        digits = r.replace("-", "").replace(".", "").lstrip("0").rstrip("0") or "0"
        exp = len(r.replace("-", "").split(".")[0]) - 1
        r = ("-" if x < 0 else "") + digits[0] + "." + (digits[1:] or "0") + f"e{exp}"
    if "e" in r:
        mant, exp = r.split("e")
        if "." not in mant:
            mant += ".0"
        return f"{mant}e{int(exp)}"
    return r


def jl_value(v) -> str:
    if v is None:
        return "nothing"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, float):
        return jl_float(v)
    return str(v)


def pct(part: int, total: int) -> str:
    """``round(part / total * 100, digits=2)`` as Julia prints it (0/0 -> NaN)."""
    if total == 0:
        return "NaN" if part == 0 else ("Inf" if part > 0 else "-Inf")
    return jl_float(float(_round2(part / total * 100.0)))


def canonical_duration(d: Optional[_dt.timedelta]) -> Optional[str]:
    """``string(Dates.canonicalize(duration))`` of a millisecond period: "1 minute, 2 seconds, 30 milliseconds"."""
    if d is None:
        return None
    ms = int(round(d.total_seconds() * 1000.0))
    if ms == 0:
        return "empty period"
    parts = []
    for name, size in (("week", 604800000), ("day", 86400000), ("hour", 3600000), ("minute", 60000), ("second", 1000),
                       ("millisecond", 1)):
        q, ms = divmod(ms, size) if ms >= 0 else (-((-ms) // size), -((-ms) % size))
        if q:
            parts.append(f"{q} {name}{'' if abs(q) == 1 else 's'}")
    return ", ".join(parts)


def _now() -> str:
    return _dt.datetime.now().strftime("%Y-%m-%d %H:%M:%S")


def _sorted_counts(stats: DemuxStats):
    # sort(collect(sample_counts), by = x -> x[2], rev = true) — ties keep a deterministic (barcode) order here
    return sorted(stats.sample_counts.items(), key=lambda kv: (-kv[1], kv[0]))


def _name(config, key) -> str:
    bc1, bc2 = key
    name = str(config.ids[bc1 - 1])
    if bc2 > 0:
        name = name + "." + str(config.ids2[bc2 - 1])
    return name


def _header_lines(stats, config, fastq_path, bc_path, fastq_path2, bc_path2, bc_complement, bc_rev, trim_side,
                  trim_side2, n_threads, duration):
    """The lines shared by the text (:71-129) and stdout (:235-298) reports."""
    L = ["BioDemuX Summary Report", "=======================", "Run Information:", f"  Date: {_now()}",
         "  Input FASTQ(s): " + fastq_path + (", " + fastq_path2 if fastq_path2 is not None else ""),
         f"  Barcode File: {bc_path}"]
    if bc_path2 is not None:
        L.append(f"  Barcode File 2: {bc_path2}")
    if n_threads is not None:
        L.append(f"  Threads: {n_threads}")
    if duration is not None:
        L.append(f"  Duration: {canonical_duration(duration)}")
    L += ["  Parameters:", f"    Max Error Rate: {jl_float(config.max_error_rate)}", f"    Min Delta: {jl_float(config.min_delta)}",
          f"    Match: {config.match}, Mismatch: {config.mismatch}, Indel: {config.indel}"]
    if config.nindel is not None:
        L.append(f"    N Indel: {config.nindel}")
    if config.classify_both:
        L.append("    Classify Both: true")
    if config.gzip_output:
        L.append("    Gzip Output: true")
    if bc_complement:
        L.append("    BC Complement: true")
    if bc_rev:
        L.append("    BC Reverse: true")
    if trim_side is not None:
        L.append(f"    Trim Side: {trim_side}")
    if trim_side2 is not None:
        L.append(f"    Trim Side 2: {trim_side2}")
    t = stats.total_reads
    L += ["", f"Total Reads: {t}", f"Matched Reads: {stats.matched_reads} ({pct(stats.matched_reads, t)}%)",
          f"Unmatched Reads: {stats.unmatched_reads} ({pct(stats.unmatched_reads, t)}%)",
          f"Ambiguous Reads: {stats.ambiguous_reads} ({pct(stats.ambiguous_reads, t)}%)", "", "Barcode Counts:"]
    for key, count in _sorted_counts(stats):
        L.append(f"{_name(config, key)}\t{count}\t{pct(count, t)}%")
    return L


def write_text_report(stats, config, output_dir, fastq_path, bc_path, fastq_path2=None, bc_path2=None, *,
                      bc_complement=False, bc_rev=False, trim_side=None, trim_side2=None, n_threads=None, duration=None):
    """reporting.jl:60-130: summary.txt, appended to (after a separator) when it exists."""
    path = os.path.join(output_dir, "summary.txt")
    append = os.path.isfile(path)
    with open(path, "a" if append else "w") as io:
        if append:
            io.write("\n==================================================\n\n")
        io.write("\n".join(_header_lines(stats, config, fastq_path, bc_path, fastq_path2, bc_path2, bc_complement, bc_rev,
                                         trim_side, trim_side2, n_threads, duration)) + "\n")


def write_stdout_report(stats, config, fastq_path, bc_path, fastq_path2=None, bc_path2=None, *, bc_complement=False,
                        bc_rev=False, trim_side=None, trim_side2=None, n_threads=None, duration=None, file=None):
    """reporting.jl:234-299."""
    print("\n".join(_header_lines(stats, config, fastq_path, bc_path, fastq_path2, bc_path2, bc_complement, bc_rev,
                                  trim_side, trim_side2, n_threads, duration)), file=file or sys.stdout)


def _key_str(k) -> str:
    if isinstance(k, tuple):
        return "(" + ", ".join(str(x) for x in k) + ")"  # string((1, 0)) == "(1, 0)"
    if isinstance(k, float):
        return jl_float(k)
    return str(k)


def dict_to_json(d: Dict) -> str:
    """reporting.jl:134-142: keys through string(), nested dictionaries recursively, no escaping.  Keys are written
    in sorted order (Julia's Dict order is a hash order nobody can rely on)."""
    items = []
    for k in sorted(d.keys()):
        v = d[k]
        items.append(f"\"{_key_str(k)}\": {dict_to_json(v) if isinstance(v, dict) else v}")
    return "{" + ", ".join(items) + "}"


def _json_null(v, quote=False) -> str:
    if v is None:
        return "null"
    return f"\"{v}\"" if quote else jl_value(v)


def write_json_report(stats, config, output_dir, fastq_path, bc_path, fastq_path2=None, bc_path2=None, *,
                      bc_complement=False, bc_rev=False, trim_side=None, trim_side2=None, n_threads=None, duration=None):
    """reporting.jl:132-232: summary.json holds a LIST of run objects; a second run appends to it."""
    run_info = f"""
    "run_info": {{
        "date": "{_now()}",
        "input_fastq": "{fastq_path}",
        "input_fastq2": {_json_null(fastq_path2, True)},
        "barcode_file": "{bc_path}",
        "barcode_file2": {_json_null(bc_path2, True)},
        "threads": {_json_null(n_threads)},
        "duration": {_json_null(canonical_duration(duration), True)},
        "parameters": {{
            "max_error_rate": {jl_float(config.max_error_rate)},
            "min_delta": {jl_float(config.min_delta)},
            "match": {config.match},
            "mismatch": {config.mismatch},
            "indel": {config.indel},
            "nindel": {_json_null(config.nindel)},
            "classify_both": {jl_value(bool(config.classify_both))},
            "gzip_output": {jl_value(bool(config.gzip_output))},
            "bc_complement": {jl_value(bool(bc_complement))},
            "bc_rev": {jl_value(bool(bc_rev))},
            "trim_side": {_json_null(trim_side)},
            "trim_side2": {_json_null(trim_side2)}
        }}
    }}
    """
    fields = ["sample_counts", "bc1_pos_counts", "bc1_len_counts", "bc1_score_counts", "bc1_per_bc_score_counts",
              "bc1_per_bc_pos_counts", "bc1_per_bc_len_counts", "bc2_pos_counts", "bc2_len_counts", "bc2_score_counts",
              "bc2_per_bc_score_counts", "bc2_per_bc_pos_counts", "bc2_per_bc_len_counts"]
    body = ",\n".join(f"        \"{f}\": {dict_to_json(getattr(stats, f))}" for f in fields)
    json_str = f"""
    {{
        {run_info},
        "total_reads": {stats.total_reads},
        "matched_reads": {stats.matched_reads},
        "unmatched_reads": {stats.unmatched_reads},
        "ambiguous_reads": {stats.ambiguous_reads},
{body}
    }}
    """
    path = os.path.join(output_dir, "summary.json")
    if os.path.isfile(path):
        with open(path) as io:
            existing = io.read().strip()
        new = existing[:-1] + ", " + json_str + "]" if existing.startswith("[") else "[" + existing + ", " + json_str + "]"
    else:
        new = "[" + json_str + "]"
    with open(path, "w") as io:
        io.write(new)


# ---- HTML (own layout) ----
_HTML_HEAD = """<!DOCTYPE html>
<html lang="en"><head><meta charset="utf-8"><title>BioDemuX Summary Report</title>
<style>
body{font-family:system-ui,Segoe UI,Helvetica,Arial,sans-serif;margin:0;background:#f5f6f8;color:#1f2328}
.container{max-width:1100px;margin:24px auto;background:#fff;border:1px solid #d8dee4;border-radius:8px;padding:24px 32px}
h1{margin-top:0} table{border-collapse:collapse;width:100%} th,td{border-bottom:1px solid #e6e8eb;padding:6px 10px;text-align:left}
.cards{display:flex;gap:16px;flex-wrap:wrap}.card{flex:1;min-width:160px;border:1px solid #d8dee4;border-radius:6px;padding:12px}
.card .v{font-size:26px;font-weight:600}.bar{background:#4c8eda;height:12px;border-radius:3px}
details{margin:8px 0} summary{cursor:pointer;font-weight:600} svg text{font-size:10px;fill:#57606a}
</style></head><body>
<div class="container"><h1>BioDemuX Summary Report</h1>
"""
_HTML_TAIL = "</div>\n</body></html>\n"


def _svg_hist(data: Dict, title: str) -> str:
    if not data:
        return f"<p>No data for {_html.escape(title)}</p>"
    keys = sorted(data)
    vals = [data[k] for k in keys]
    top = max(vals)
    W, H, left, bottom = 560, 220, 46, 34
    step = (W - left) / len(keys)
    bars = []
    for i, (k, v) in enumerate(zip(keys, vals)):
        h = (H - bottom - 10) * v / top if top else 0
        x = left + i * step
        bars.append(f'<rect x="{x + step * 0.1:.1f}" y="{H - bottom - h:.1f}" width="{step * 0.8:.1f}" height="{h:.1f}" fill="#4c8eda">'
                    f"<title>{_key_str(k)}: {v}</title></rect>")
        if len(keys) <= 24 or i % max(1, len(keys) // 12) == 0:
            bars.append(f'<text x="{x + step / 2:.1f}" y="{H - bottom + 12}" text-anchor="middle">{_key_str(k)}</text>')
    axis = (f'<line x1="{left}" y1="{H - bottom}" x2="{W}" y2="{H - bottom}" stroke="#8c959f"/>'
            f'<line x1="{left}" y1="8" x2="{left}" y2="{H - bottom}" stroke="#8c959f"/>'
            f'<text x="{left - 4}" y="14" text-anchor="end">{top}</text><text x="{left - 4}" y="{H - bottom}" text-anchor="end">0</text>')
    return (f'<figure><figcaption>{_html.escape(title)}</figcaption>'
            f'<svg width="{W}" height="{H}" viewBox="0 0 {W} {H}">{axis}{"".join(bars)}</svg></figure>')


def write_html_report(stats, config, output_dir, fastq_path, bc_path, fastq_path2=None, bc_path2=None, *,
                      bc_complement=False, bc_rev=False, trim_side=None, trim_side2=None, n_threads=None, duration=None):
    """reporting.jl:301-565: summary.html; a later run adds its section before </body>."""
    e = _html.escape
    params = [f"Max Error Rate: {jl_float(config.max_error_rate)}", f"Min Delta: {jl_float(config.min_delta)}",
              f"Match: {config.match}, Mismatch: {config.mismatch}, Indel: {config.indel}"]
    if config.nindel is not None:
        params.append(f"N Indel: {config.nindel}")
    for flag, text in ((config.classify_both, "Classify Both: true"), (config.gzip_output, "Gzip Output: true"),
                       (bc_complement, "BC Complement: true"), (bc_rev, "BC Reverse: true")):
        if flag:
            params.append(text)
    if trim_side is not None:
        params.append(f"Trim Side: {trim_side}")
    if trim_side2 is not None:
        params.append(f"Trim Side 2: {trim_side2}")
    t = stats.total_reads
    info = [f"<li><strong>Date:</strong> {_now()}</li>",
            f"<li><strong>Input FASTQ(s):</strong> {e(fastq_path)}{', ' + e(fastq_path2) if fastq_path2 is not None else ''}</li>",
            f"<li><strong>Barcode File:</strong> {e(bc_path)}</li>"]
    if bc_path2 is not None:
        info.append(f"<li><strong>Barcode File 2:</strong> {e(bc_path2)}</li>")
    if n_threads is not None:
        info.append(f"<li><strong>Threads:</strong> {n_threads}</li>")
    if duration is not None:
        info.append(f"<li><strong>Duration:</strong> {canonical_duration(duration)}</li>")
    info.append("<li><strong>Parameters:</strong><ul>" + "".join(f"<li>{e(p)}</li>" for p in params) + "</ul></li>")
    cards = "".join(f'<div class="card"><div>{name}</div><div class="v">{val}</div><div>{extra}</div></div>' for name, val, extra in (
        ("Total Reads", t, ""), ("Matched", stats.matched_reads, pct(stats.matched_reads, t) + "%"),
        ("Unmatched", stats.unmatched_reads, pct(stats.unmatched_reads, t) + "%"),
        ("Ambiguous", stats.ambiguous_reads, pct(stats.ambiguous_reads, t) + "%")))
    counts = _sorted_counts(stats)
    top = counts[0][1] if counts else 0
    rows = "".join(f"<tr><td>{e(_name(config, k))}</td><td>{c}</td><td>{pct(c, t)}%</td>"
                   f'<td style="width:40%"><div class="bar" style="width:{(c / top * 100 if top else 0):.1f}%"></div></td></tr>'
                   for k, c in counts)
    charts = []
    for tag, title in (("bc1", "Barcode 1"), ("bc2", "Barcode 2")):
        if not getattr(stats, f"{tag}_pos_counts"):
            continue
        charts.append(f"<h3>{title}: global</h3>" + _svg_hist(getattr(stats, f"{tag}_score_counts"), "Score")
                      + _svg_hist(getattr(stats, f"{tag}_pos_counts"), "Start Position") + _svg_hist(getattr(stats, f"{tag}_len_counts"), "Length"))
        ids = config.ids if tag == "bc1" else config.ids2
        for b in sorted(getattr(stats, f"{tag}_per_bc_pos_counts")):
            charts.append(f"<details><summary>Stats for {e(str(ids[b - 1]))} ({title})</summary>"
                          + _svg_hist(getattr(stats, f"{tag}_per_bc_score_counts").get(b, {}), "Score")
                          + _svg_hist(getattr(stats, f"{tag}_per_bc_pos_counts").get(b, {}), "Start Position")
                          + _svg_hist(getattr(stats, f"{tag}_per_bc_len_counts").get(b, {}), "Length") + "</details>")
    body = (f'<section class="run-section"><h2>Run Information</h2><ul>\n' + "\n".join(info) + "\n</ul>"
            f'<div class="cards">{cards}</div><h2>Barcode Statistics</h2>'
            f"<table><tr><th>Barcode</th><th>Count</th><th>Percentage</th><th>Distribution</th></tr>{rows}</table>"
            f"<h2>Detailed Distributions</h2>{''.join(charts)}</section><hr>\n")
    path = os.path.join(output_dir, "summary.html")
    if os.path.isfile(path):
        with open(path) as io:
            existing = io.read()
        new = existing.replace("</body>", '<div class="container">\n' + body + "</div>\n</body>", 1)
        with open(path, "w") as io:
            io.write(new)
    else:
        with open(path, "w") as io:
            io.write(_HTML_HEAD + body + _HTML_TAIL)


def generate_summary_report(stats: DemuxStats, config, output_dir: str, fastq_path: str, bc_path: str,
                            fastq_path2: Optional[str] = None, bc_path2: Optional[str] = None, **kw) -> None:
    """reporting.jl:567-577."""
    fmt = str(config.summary_format).lstrip(":")
    if fmt == "json":
        write_json_report(stats, config, output_dir, fastq_path, bc_path, fastq_path2, bc_path2, **kw)
    elif fmt == "html":
        write_html_report(stats, config, output_dir, fastq_path, bc_path, fastq_path2, bc_path2, **kw)
    elif fmt == "stdout":
        write_stdout_report(stats, config, fastq_path, bc_path, fastq_path2, bc_path2, **kw)
    else:
        write_text_report(stats, config, output_dir, fastq_path, bc_path, fastq_path2, bc_path2, **kw)
