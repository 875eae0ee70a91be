"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one bench.py command.

Unit and correction as MI355X_MICROARCH.md (HBM) prescribes for gfx950: the counters are in KiB, FETCH_SIZE tallies
128-B read requests at 64 B, so  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Output: one JSON with, per kernel
family, launches and average bytes per launch -- bench.py reports it as roofline.traffic for the dominant kernel.

usage: python tools/pmc_traffic.py <dir with FETCH pass> <dir with WRITE pass> <out.json> [workload]
"""
import csv, glob, json, re, sys


def family(name):
    m = re.search(r"conv_gemm_kernel<\(int\)\d+, \(int\)(\d+), \(int\)(\d+)", name) or re.search(r"conv_gemm_kernel<\d+, (\d+), (\d+)", name)
    if m:
        return f"conv_gemm_kernel<{m.group(1)}x{m.group(2)}>"
    m = re.search(r"conv_c64_kernel<(?:\(int\))?\d+, (?:\(int\))?(\d+)>", name)
    if m:
        return f"conv_c64_kernel<{m.group(1)}>"            # same labels as dsr_conv_kernel_name()
    if "conv_wgrad_tile_kernel" in name:
        return "conv_wgrad_tile_kernel<1x1>"
    for k in ("conv_wgrad_dma_batch_kernel", "conv_wgrad_dma_s2_kernel", "conv_wgrad_dma_kernel", "conv_smalln_kernel", "conv_wgrad_taps_kernel", "conv_wgrad_kernel"):
        if k in name:
            return k
    return re.sub(r"<.*", "", name.split("(")[0]).strip()


def collect(directory, counter):
    tot, cnt = {}, {}
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = family(row["Kernel_Name"])
                tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
                cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "gan_x4"
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    fams = {}
    for k in sorted(set(ft) | set(wt)):
        n = fc.get(k) or wc.get(k)
        rd = 2.0 * ft.get(k, 0.0) * 1024 / max(fc.get(k, 1), 1)
        wr = wt.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1)
        fams[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr}
    json.dump({"workload": workload, "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged per launch",
               "families": fams}, open(out, "w"), indent=1)
    for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * (kv[1]["launches"] or 0))[:12]:
        print(f"{k:36s} n={v['launches']:5d}  {v['bytes_per_launch'] / 1e6:9.2f} MB/launch")


if __name__ == "__main__":
    main()
