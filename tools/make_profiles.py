#!/usr/bin/env python3
"""Turn one round's raw measurement files (gpurun_out/, written by tools/profile_round.sh <tag> on the GPU box) into the
committed summaries under profiles/.

Inputs:
  gpurun_out/<tag>_bench_final.log        python bench.py                               (last line = the JSON)
  gpurun_out/<tag>_f_kernel_stats.csv     rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline
  gpurun_out/pmc_<tag>_{fetch,write,sq1,sq2}.csv   tools/pmc_summary.py over rocprofv3 --pmc passes of
                                          python3 bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline
  gpurun_out/<tag>_config3.json, _config5.json, _gn_kernel_stats.csv, _pmc_gn.csv   tools/run_config3.py (plain, traced, counted)
usage: make_profiles.py <round tag, e.g. r02>"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"

# rocprofv3's kernel names -> the names bench.py reports (asl_stage_times)
NAMES = {"void k_seg_points<4096, 2048, 4, 1>": "k_seg_points", "void k_seg_points<8192, 4352, 1, 2>": "k_seg_points<dense>",
         "void k_decimate2_tiles<3>": "k_decimate_minmax", "void k_decimate_rest<3>": "k_decimate_rest",
         "void k_decimate_minmax<3>": "k_decimate_minmax<generic>", "void k_decode<3>": "k_decode", "void k_refine<3>": "k_refine",
         "void k_fit_quads<64, true, 2>": "k_fit_quads<0>", "void k_fit_quads<128, true, 2>": "k_fit_quads<1>",
         "void k_fit_quads<256, true, 2>": "k_fit_quads<2>", "void k_fit_quads<256, true, 4>": "k_fit_quads<3>",
         "void k_fit_quads<256, false, 0>": "k_fit_quads<4>"}


def short(k):
    k = k.strip('"')
    k = re.sub(r"\(.*$", "", k)          # drop the argument list
    return NAMES.get(k, k)


def table(path):
    rows = list(csv.reader(open(path)))
    hdr = rows[0]
    return hdr, {r[0]: dict(zip(hdr[1:], [float(x) for x in r[1:]])) for r in rows[1:]}


def ours(k):
    return not (k.startswith("__amd") or "at::native" in k)


bench = json.loads(open(os.path.join(G, tag + "_bench_final.log")).read().strip().splitlines()[-1])
B = bench["config"]["batch_frames"]
json.dump(bench, open(os.path.join(P, tag + "_bench_default.json"), "w"))

dst = os.path.join(P, tag + "_f_kernel_stats.csv")
shutil.copy(os.path.join(G, tag + "_f_kernel_stats.csv"), dst)
body = open(dst).read()
open(dst, "w").write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (B=%d, pipeline 2, 20 steps + 5 warm-up; "
                     "only warm-up/timed launches overlap with the other workspace)\n" % B + body)

_, f = table(os.path.join(G, "pmc_%s_fetch.csv" % tag))
_, w = table(os.path.join(G, "pmc_%s_write.csv" % tag))
bpl = {}
with open(os.path.join(P, tag + "_g_pmc_fetch_write_per_kernel.csv"), "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE (second run: --pmc WRITE_SIZE) -- python3 bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline\n")
    o.write("# one launch = %d frames of 1280x720 BGR.  Counter unit: KB, average per launch.  gfx950: FETCH_SIZE counts 64 B per 128-B request on "
            "wide (16 B/lane) coalesced streams (x2 to compare with bytes); byte/dword gathers are uncalibrated, values are raw.\n" % B)
    kd = [k for k in f if short(k) == "k_decimate_minmax"]
    if kd:
        known = 360 * 1280 * 3 * B / 1024.0  # KB: every second row of the 720p BGR frames, read exactly once
        o.write("# calibration on a known byte count (MI355X_MICROARCH.md, HBM): k_decimate_minmax reads %.0f KB per launch by construction "
                "(every second row of the input, once); FETCH_SIZE reports %.0f KB: x%.2f for its 16 + 4 + 1 byte loads at a 24-byte stride "
                "(the guide's x2 holds for pure 16 B/lane streams); the other kernels' access widths are uncalibrated.\n"
                % (known, f[kd[0]]["FETCH_SIZE"], known / f[kd[0]]["FETCH_SIZE"]))
    o.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,raw_bytes_per_frame(fetch+write)\n")
    for k in sorted(f):
        if not ours(k):
            continue
        fv, wv = f[k]["FETCH_SIZE"], w.get(k, {}).get("WRITE_SIZE", 0.0)
        nm = short(k)
        bpl[nm] = (fv + wv) * 1024.0
        o.write("%s,%d,%.1f,%.1f,%.0f\n" % (nm, int(f[k]["launches"]), fv, wv, (fv + wv) * 1024 / B))
json.dump({"source": "profiles/%s_g_pmc_fetch_write_per_kernel.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, B=%d)" % (tag, B),
           "note": "raw FETCH_SIZE+WRITE_SIZE bytes per launch; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950",
           "batch_frames": B, "bytes_per_launch": bpl}, open(os.path.join(P, tag + "_traffic.json"), "w"), indent=1)

h1, a = table(os.path.join(G, "pmc_%s_sq1.csv" % tag))
h2, b = table(os.path.join(G, "pmc_%s_sq2.csv" % tag))
c1, c2 = h1[2:], h2[2:]
iso = bench["kernel_ms_per_batch_isolated"]
with open(os.path.join(P, tag + "_h_sq_counters_per_kernel.csv"), "w") as o:
    o.write("# rocprofv3 --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline ; B=%d frames "
            "per launch, averages per launch\n" % B)
    o.write("# valu_floor_ms = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz): the time the launch would take if the vector ALUs never idled; "
            "isolated_ms from the same build (bench.py kernel_ms_per_batch_isolated)\n")
    o.write("kernel," + ",".join(c1 + c2) + ",valu_floor_ms,isolated_ms,valu_busy_frac\n")
    for k in sorted(a):
        if not ours(k):
            continue
        nm = short(k)
        v = a[k]["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9) * 1e3
        t = iso.get(nm)
        o.write(nm + "," + ",".join("%.0f" % a[k][c] for c in c1) + "," + ",".join("%.0f" % b.get(k, {}).get(c, 0) for c in c2) +
                ",%.3f,%s,%s\n" % (v, ("%.3f" % t) if t else "", ("%.2f" % (v / t)) if t else ""))
print("value", bench["value"], "stage", bench["stage_threshold_segmentation"], "traffic/frame", sum(bpl.values()) / B)

sq3 = os.path.join(G, "pmc_%s_sq3.csv" % tag)
if os.path.exists(sq3):
    h3, m3 = table(sq3)
    with open(os.path.join(P, tag + "_i_valu_mix_per_kernel.csv"), "w") as o:
        o.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -- python3 bench.py --steps 2 "
                "--warmup 1 --pipeline 1 --no-cpu-baseline ; wave-instructions per launch (B=%d frames), shares of SQ_INSTS_VALU\n" % B)
        o.write("kernel,SQ_INSTS_VALU,add_f64,mul_f64,fma_f64,trans_f64,int32,cvt\n")
        for k in sorted(m3):
            v = m3[k].get("SQ_INSTS_VALU", 0.0)
            if not ours(k) or v <= 0:
                continue
            o.write("%s,%.0f,%s\n" % (short(k), v, ",".join("%.3f" % (m3[k].get(c, 0.0) / v) for c in (
                "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT"))))

# the bench line was printed before this round's traffic file existed: fill the traffic fields from the same passes
seg_members = list(bench["roofline"].get("members_ms", {}))
if bench["roofline"].get("traffic") is None and seg_members:
    src_note = "profiles/%s_g_pmc_fetch_write_per_kernel.csv (same build, counted right after this run)" % tag
    bench["roofline"]["traffic"] = sum(bpl.get(k, 0.0) for k in seg_members)
    bench["roofline"]["traffic_source"] = src_note
    dk = bench["roofline"].get("dominant_kernel") or {}
    if dk.get("traffic") is None and dk.get("kernel") in bpl:
        dk["traffic"] = bpl[dk["kernel"]]
        dk["traffic_source"] = src_note
    json.dump(bench, open(os.path.join(P, tag + "_bench_default.json"), "w"))

for src, dst_name in ((tag + "_config3.json", tag + "_config3_detect_pnp_gn.json"), (tag + "_config5.json", tag + "_config5_single_gpu_detect_pnp_gn.json"),
                      (tag + "_gn_kernel_stats.csv", tag + "_gn_kernel_stats.csv"), (tag + "_pmc_gn.csv", tag + "_gn_mfma_counters.csv"),
                      (tag + "_exchange_n1.json", tag + "_bench_exchange_n1.json"), (tag + "_rehearse_gpus2.json", tag + "_bench_rehearse_gpus2_gloo.json"),
                      (tag + "_bench_configs2.json", tag + "_bench_configs2.json"), (tag + "_bench_configs4_n1.json", tag + "_bench_configs4_n1.json"),
                      (tag + "_bench_configs4_rehearse_gpus2.json", tag + "_bench_configs4_rehearse_gpus2_gloo.json"),
                      (tag + "_marker_trace.csv", tag + "_marker_trace.csv")):
    if os.path.exists(os.path.join(G, src)):
        if src.endswith(".json"):  # a library may have written to stdout before the line (gloo's connection notice)
            lines = [ln for ln in open(os.path.join(G, src)) if ln.startswith("{")]
            open(os.path.join(P, dst_name), "w").write(lines[-1] if lines else open(os.path.join(G, src)).read())
        else:
            shutil.copy(os.path.join(G, src), os.path.join(P, dst_name))
