"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; each run with --kernel-trace --output-format csv) into
profiles/<round>_pmc_traffic.json: HBM-side bytes per launch for every kernel of the hot path.

    bytes = FETCH_SIZE * 1024 * 2  +  WRITE_SIZE * 1024
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads, hence the x2:
MI355X_MICROARCH.md, HBM / rocprofv3 section.)  usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] += float(row["Counter_Value"])
            n[k] += 1
    return tot, n


AMODE = {"0": "conv1x1", "1": "conv3x3", "2": "stem_conv", "3": "dcn3x3"}
DT = {"DF16b": "bf16", "DF16_": "f16", "f": "f32", "NS_7f16x2_tE": "f16x2"}


DEM = {"float": "f32", "__bf16": "bf16", "_Float16": "f16", "ocrvi::f16x2_t": "f16x2"}


def tag_of(name):
    """Kernel name (mangled or demangled, as rocprofv3 prints it) -> the tag bench.py's profiler uses for that kernel."""
    m = re.search(r"gemm_ring_kernel<(float|__bf16|_Float16|ocrvi::f16x2_t)", name)
    if m:
        return "gemm_ring_" + DEM[m.group(1)]
    m = re.search(r"conv_gemm_kernel<(float|__bf16|_Float16|ocrvi::f16x2_t), (\d), (\d+), (\d+)", name)
    if m:
        return f"{AMODE.get(m.group(2), 'conv')}_{m.group(3)}x{m.group(4)}_{DEM[m.group(1)]}"
    m = re.search(r"attention(16)?_kernel<(float|__bf16|_Float16|ocrvi::f16x2_t)", name)
    if m:
        return "attention_hd32_" + DEM[m.group(2)]
    m = re.search(r"dcn_pipe_kernel<(float|__bf16|_Float16|ocrvi::f16x2_t), (\d+)(?:, (?:true|false))?>", name)
    if m:
        return f"dcn3x3_pipe128x{m.group(2)}_{DEM[m.group(1)]}"
    m = re.search(r"offs_conv_kernel<(float|__bf16|_Float16|ocrvi::f16x2_t)", name)
    if m:
        return "dcn_offset_conv3x3_128x32_" + DEM[m.group(1)]
    m = re.search(r"offs_conv_kernelI(DF16b|DF16_|f|NS_7f16x2_tE)", name)
    if m:
        return "dcn_offset_conv3x3_128x32_" + DT[m.group(1)]
    m = re.search(r"mlp_fused_kernel<(__bf16|_Float16), (\d+)", name)
    if m:
        return f"mlp_fused_d{m.group(2)}_{DEM[m.group(1)]}"
    m = re.search(r"gconv32_kernel<(__bf16|_Float16)", name)
    if m:
        return "gconv3x3_128x32_" + DEM[m.group(1)]
    m = re.search(r"gemm_ring_kernelI(DF16b|DF16_|f|NS_7f16x2_tE)", name)
    if m:
        return "gemm_ring_" + DT.get(m.group(1), m.group(1))
    m = re.search(r"conv_gemm_kernelI(DF16b|DF16_|f|NS_7f16x2_tE)Li(\d)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"{AMODE.get(m.group(2), 'conv')}_{m.group(3)}x{m.group(4)}_{DT[m.group(1)]}"
    m = re.search(r"mlp_fused_kernelI(DF16b|DF16_)Li(\d+)E", name)
    if m:
        return f"mlp_fused_d{m.group(2)}_{DT[m.group(1)]}"
    m = re.search(r"dcn_pipe_kernelI(DF16b|DF16_|f|NS_7f16x2_tE)Li(\d+)E", name)
    if m:
        return f"dcn3x3_pipe128x{m.group(2)}_{DT[m.group(1)]}"
    m = re.search(r"attention(16)?_kernelI(DF16b|DF16_|f|NS_7f16x2_tE)", name)
    if m:
        return "attention_hd32_" + DT[m.group(2)]
    m = re.search(r"gconv32_kernelI(DF16b|DF16_)", name)
    if m:
        return "gconv3x3_128x32_" + DT[m.group(1)]
    if "stem_pool_kernel" in name:
        return "stem_pool_f16x2"
    m = re.search(r"mlp_x2_kernel<(\d+)|mlp_x2_kernelILi(\d+)E", name)
    if m:
        return f"mlp_fused_d{m.group(1) or m.group(2)}_f16x2"
    m = re.search(r"ocrvi::?(\w+?)_kernel|5ocrvi\d+(\w+?)_kernel", name)
    return (m.group(1) or m.group(2)) if m else name[:60]


if __name__ == "__main__":
    fd, wd, out = sys.argv[1:4]
    ft, fn = collect(fd, "FETCH_SIZE")
    wt, wn = collect(wd, "WRITE_SIZE")
    agg = defaultdict(lambda: dict(dispatches=0, fetch=0.0, write=0.0, mangled=[]))
    for k in ft:
        a = agg[tag_of(k)]
        a["dispatches"] += fn[k]
        a["fetch"] += ft[k] * 1024 * 2
        a["write"] += wt.get(k, 0.0) * 1024
        a["mangled"].append(k)
    kernels = {}
    for t, a in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
        if not a["dispatches"]:
            continue
        kernels[t] = {"mangled": a["mangled"], "dispatches": a["dispatches"], "fetch_bytes_per_launch": round(a["fetch"] / a["dispatches"]),
                      "write_bytes_per_launch": round(a["write"] / a["dispatches"]),
                      "traffic_bytes_per_launch": round((a["fetch"] + a["write"]) / a["dispatches"])}
    json.dump({"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) over `python3 bench.py "
                          "--steps 1 --warmup 1 --no-graph --no-prof --no-cpu-baseline --no-parity-check` (every mode of the default run); per-launch averages over all dispatches of the kernels that share a "
                          "tag; bytes = FETCH_SIZE*1024*2 (gfx950 counts 64 B per 128-B request on wide coalesced reads: MI355X_MICROARCH.md, HBM) + "
                          "WRITE_SIZE*1024.  Infinity-Cache hits are included in these fabric-side counters.  Made by tools/pmc_traffic.py.",
               "kernels": kernels}, open(out, "w"), indent=1)
    for t, k in list(kernels.items())[:12]:
        print(f"{t:32s} x{k['dispatches']:5d}  {k['traffic_bytes_per_launch']/1e6:9.1f} MB/launch")
