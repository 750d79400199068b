"""Write the number tables of profiles/README_rNN.md and DESIGN.md FROM the committed summaries of round NN -- no kernel
number in tracked prose is typed by hand.

    python tools/render_numbers.py r04            # rewrite the generated blocks
    python tools/render_numbers.py r04 --check    # exit 1 if a committed block differs from what the summaries say

Inputs (profiles/):  <r>_kernel_stats_bench_single_stream.csv, <r>_kernel_stats_bench_default.csv (rocprofv3
--kernel-trace --stats), <r>_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes), <r>_pmc_sq.json (SQ counters),
<r>_isa_mix.txt (tools/isa_mix.py over csrc/*.hip), <r>_bench_default.json (the line of `python bench.py`).
A generated block sits between `<!-- BEGIN GENERATED <name> (tools/render_numbers.py) -->` and `<!-- END GENERATED <name> -->`
in profiles/README_<r>.md and DESIGN.md; everything outside the markers is prose and is left alone."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(name):
    """`k_name` or `k_name<template arguments>` out of rocprofv3's full signature."""
    m = re.search(r"\(anonymous namespace\)::(k_[A-Za-z0-9_]+)(<.*?>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def kernel_stats(path):
    """-> {kernel: (calls, avg_us)} of our kernels."""
    out = {}
    for r in csv.DictReader(open(path)):
        if "(anonymous namespace)::k_" not in r["Name"]:
            continue
        out[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def isa_table(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        f = line.split()
        if len(f) >= 12:
            out[re.sub(r"<L[bi].*", lambda m: m.group(0), f[0])] = dict(VALU=int(f[1]), f64=int(f[2]), VGPR=int(f[7]), waves=int(f[9]),
                                                                      lds=int(f[10]), scratch=int(f[11]))
    return out


def isa_lookup(isa, kernel):
    base = kernel.split("<")[0]
    hits = [v for k, v in isa.items() if k.split("<")[0] == base]
    if not hits:
        return None
    return max(hits, key=lambda v: v["VGPR"])           # (template instances: the heaviest one)


def fmt_mb(b):
    return "-" if b is None else "%.0f" % (b / 1e6)


def render(r):
    ks = kernel_stats(os.path.join(P, r + "_kernel_stats_bench_single_stream.csv"))
    ksd = kernel_stats(os.path.join(P, r + "_kernel_stats_bench_default.csv"))
    pm = json.load(open(os.path.join(P, r + "_pmc_traffic.json")))
    sq = json.load(open(os.path.join(P, r + "_pmc_sq.json")))
    isa = isa_table(os.path.join(P, r + "_isa_mix.txt"))
    bj = json.load(open(os.path.join(P, r + "_bench_default.json")))
    passes = ks["k_bp_bin"][0]                             # one launch of the binning pass per pass
    blocks = {}

    # ---- kernels of one single-stream pass
    rows = []
    for k, (calls, avg) in ks.items():
        per_pass = calls / passes
        if per_pass < 0.99:                                # one-off kernels (table init, mask packing)
            continue
        t = pm["kernels"].get(k) or next((v for n, v in pm["kernels"].items() if n.split("<")[0] == k.split("<")[0]), None)
        s = sq.get(k.split("<")[0])
        i = isa_lookup(isa, k)
        rows.append((avg * per_pass, k, per_pass, avg, t, s, i))
    rows.sort(reverse=True)
    total = sum(x[0] for x in rows)
    lines = ["| kernel | launches / pass | avg µs | µs / pass | fetched MB | written MB | VALU / wave | wait_any | VGPR | waves / SIMD | scratch B |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for tp, k, pp, avg, t, s, i in rows:
        lines.append("| `%s` | %g | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
            k, round(pp, 2), avg, tp, fmt_mb(t and t.get("fetch_bytes")), fmt_mb(t and t.get("write_bytes")),
            "-" if not s else "%.0f" % s["valu_per_wave"], "-" if not s else "%.2f" % s["wait_any_frac"],
            "-" if not i else i["VGPR"], "-" if not i else i["waves"], "-" if not i else i["scratch"]))
    lines.append("| **sum** | | | **%.0f** | | | | | | | |" % total)
    blocks["kernel_table"] = "\n".join(lines)

    # ---- radius filter stage from the kernel durations
    rf = [k for k in ks if k.split("<")[0] in ("k_rf_stream", "k_rf_pair", "k_rf_resolve", "k_rf_ranges", "k_seg_compact_short")]
    rf_us = sum(ks[k][1] * ks[k][0] / passes for k in rf)
    rfb = bj["radius_filter_roofline"]
    alg = rfb["stage"]["alg_bytes_per_launch"]
    lines = ["| | µs | |", "|---|---|---|"]
    for k in sorted(rf, key=lambda k: -ks[k][1]):
        lines.append("| `%s` | %.1f | |" % (k, ks[k][1]))
    lines.append("| **stage** (kernel durations, `%s_kernel_stats_bench_single_stream.csv`) | **%.1f** | %d points x 21 B = %.1f MB -> %.0f GB/s = **%.3f** of 8 TB/s |"
                 % (r, rf_us, rfb["points_per_launch"], alg / 1e6, alg / rf_us / 1e3, alg / rf_us / 1e3 / 8000.0))
    lines.append("| stage, HIP events around the phase calls (`bench.py`, memsets and launch gaps inside) | %.1f | %.3f |"
                 % (rfb["stage"]["avg_ms"] * 1e3, rfb["stage"]["frac"]))
    bp = rfb.get("big_pool")
    if bp and "avg_ms" in bp:
        lines.append("| the same stage on %d frames in one launch (shadow %.0f MB > 256 MiB memory-side cache; HIP events) | %.1f | %d points -> %.0f GB/s = **%.3f** |"
                     % (bp["frames"], bp["shadow_MB"], bp["avg_ms"] * 1e3, bp["points_per_launch"], bp["achieved"], bp["frac"]))
    blocks["radius_filter"] = "\n".join(lines)

    # ---- stages of the bench line
    lines = ["| stage (HIP events, single-stream passes) | ms / pass | algorithmic MB / pass | fraction of 8 TB/s |", "|---|---|---|---|"]
    for row in bj["kernels"]:
        ab = row.get("alg_bytes_per_launch")
        lines.append("| %s | %.3f | %s | %s |" % (row["stage"], row.get("ms_per_pass", row["avg_ms"]),
                                                  "-" if ab is None else "%.1f" % (ab * row.get("launches_per_pass", 1) / 1e6),
                                                  "-" if "frac_of_peak" not in row else "%.4f" % row["frac_of_peak"]))
    blocks["stage_table"] = "\n".join(lines)

    # ---- headline
    ro, pa, mp = bj["roofline"], bj.get("pass", {}), bj.get("mask_packing", {})
    par = bj.get("parity", {})
    dom = ro["kernel"]
    dk = ks.get(dom, (0, 0.0))
    head = [
        "* `python bench.py` (2 streams x 32-frame chunks): **%.0f frames/s**, %.3f ms per %d frames (`%s_bench_default.json`)."
        % (bj["value"], bj["ms_per_step"], bj["config"]["frames_per_gpu_per_step"], r),
        "* Dominant kernel `%s`: %.1f µs per %d views in the rocprofv3 summary (%.1f µs by HIP events in that run); "
        "%.1f MB algorithmic -> %.0f GB/s = **%.4f** of 8 TB/s (bench line: %.5f); PMC %s MB fetched + %s MB written per launch."
        % (dom, dk[1], 384, ro["avg_ms"] * 1e3, ro["alg_bytes_per_launch"] / 1e6, ro["alg_bytes_per_launch"] / max(dk[1], 1e-9) / 1e3,
           ro["alg_bytes_per_launch"] / max(dk[1], 1e-9) / 1e3 / 8000.0, ro["frac"],
           fmt_mb(pm["kernels"].get(dom, {}).get("fetch_bytes")), fmt_mb(pm["kernels"].get(dom, {}).get("write_bytes"))),
        "* Whole single-stream pass: sum of kernel durations %.0f µs (CSV); bench `pass`: %.3f ms, %.2f GB algorithmic = **%.4f** of 8 TB/s."
        % (total, pa.get("ms", 0.0), pa.get("alg_bytes", 0) / 1e9, pa.get("frac", 0.0)),
    ]
    if mp:
        head.append("* Mask packing (`dfu3d_pack_masks` on the step's uint8 planes): %.3f ms per step -> **%.0f frames/s** pack-inclusive."
                    % (mp["pack_ms_per_step"], mp["value_pack_inclusive"]))
    cb, ca = bj.get("cpu_baseline"), bj.get("cpu_baseline_allcores")
    if cb:
        head.append("* CPU baseline (oracle, 1 thread): %.3f frames/s%s." % (
            cb["value"], "" if not ca else "; %d processes (the job's CPU quota): %.2f frames/s" % (ca["cores"], ca["value"])))
    if par:
        head.append("* `parity`: %d rows of %d frames, keys and classes %s, largest difference %.1e m / %.1e rad."
                    % (par.get("rows_gpu", 0), par.get("frames", 0), "equal" if par.get("rows_equal") else "DIFFERENT",
                       par.get("max_abs_centre_dims_m", float("nan")), par.get("max_abs_yaw_rad", float("nan"))))
    d2 = ksd.get(dom)
    if d2:
        head.append("* The same kernel under the default command's two overlapping streams (192 views per launch, contended): %.1f µs."
                    % d2[1])
    blocks["headline"] = "\n".join(head)

    # ---- kernels that use scratch memory
    sc = sorted((k, v["scratch"], v["VGPR"]) for k, v in isa.items() if v["scratch"] > 0)
    blocks["scratch"] = ("Kernels with scratch memory (`%s_isa_mix.txt`): " % r) + (
        "; ".join("`%s` %d B (%d VGPRs)" % x for x in sc) if sc else "none") + "."
    return blocks


def apply(path, blocks, check):
    text = open(path).read()
    changed = False
    for name, body in blocks.items():
        pat = re.compile(r"(<!-- BEGIN GENERATED %s \(tools/render_numbers\.py\) -->\n)(.*?)(<!-- END GENERATED %s -->)" % (name, name), re.S)
        m = pat.search(text)
        if not m:
            continue
        if m.group(2) != body + "\n":
            changed = True
            text = text[:m.start(2)] + body + "\n" + text[m.end(2):]
    if changed and not check:
        open(path, "w").write(text)
    return changed


def main():
    r = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r04"
    check = "--check" in sys.argv
    blocks = render(r)
    stale = [p for p in (os.path.join(P, "README_%s.md" % r), os.path.join(ROOT, "DESIGN.md")) if os.path.exists(p) and apply(p, blocks, check)]
    if check:
        if stale:
            print("generated blocks out of date in: " + ", ".join(os.path.relpath(p, ROOT) for p in stale))
            sys.exit(1)
        print("generated blocks match the committed summaries")
    else:
        print("rewrote: " + (", ".join(os.path.relpath(p, ROOT) for p in stale) or "nothing (up to date)"))


if __name__ == "__main__":
    main()
