"""Summarise tools/pmc_bench.sh output: calibration factors and corrected HBM traffic of the conv kernels."""
import csv, collections, json, sys, os
out = sys.argv[1]

def load(path, counter):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg

def first(agg, key):
    for k, v in agg.items():
        if key in k:
            return v
    return []

cf = load(f"{out}/cal_fetch/f_counter_collection.csv", "FETCH_SIZE")
cw = load(f"{out}/cal_write/w_counter_collection.csv", "WRITE_SIZE")
n_vec = 8 * 64 * 724 * 724 * 4
n_sc = 8 * 64 * 723 * 723 * 4
# FETCH_SIZE / WRITE_SIZE are reported in KiB
cal = {
    "fetch_vec4": n_vec / (first(cf, "fba_vec4")[-1] * 1024), "fetch_scalar": n_sc / (first(cf, "fba_scalar")[-1] * 1024),
    "write_vec4": n_vec / (first(cw, "fba_vec4")[-1] * 1024), "write_scalar": n_sc / (first(cw, "fba_scalar")[-1] * 1024),
}
bf = load(f"{out}/fetch/f_counter_collection.csv", "FETCH_SIZE")
bw = load(f"{out}/write/w_counter_collection.csv", "WRITE_SIZE")
FAMILIES = {"wino_f2": ("conv_wino_kernel", "conv_wino_ro_kernel", "conv_wino_rod_kernel", "conv_wino_rs_kernel"), "wino_f4": ("wino4_gemm_kernel", "wino4_input_kernel"),
            "wino_f4_fused": ("conv_wino4f_kernel", "conv_wino4f_groups_kernel"),
            "pipe": ("conv_pipe",), "igemm": ("conv_igemm",), "smallmap": ("conv_smallmap",), "bf16": ("conv_bf16_kernel",), "bf16_rv": ("conv_bf16_rv_kernel",), "bf16_dg": ("conv_bf16_dg_kernel",)}
ALL = tuple(t for ts in FAMILIES.values() for t in ts)
conv_f = [v for k, vs in bf.items() if any(t in k for t in ALL) for v in vs]
conv_w = [v for k, vs in bw.items() if any(t in k for t in ALL) for v in vs]
# bench ran 1 warm-up + 1 timed step: half of the launches belong to one step (an F(4x4) conv launch = two dispatches: counted once)
n_input = sum(len(vs) for k, vs in bf.items() if "wino4_input_kernel" in k)
launches = (len(conv_f) - n_input) // 2
# conv input patches are 4-byte loads (scalar factor), weights 16-byte loads; outputs are 4-byte stores
fetch_bytes = sum(conv_f) * 1024 * cal["fetch_scalar"] / 2
write_bytes = sum(conv_w) * 1024 * cal["write_scalar"] / 2
split = {}
for fam, keys in FAMILIES.items():
    ff = [v for k, vs in bf.items() if any(t in k for t in keys) for v in vs]
    ww = [v for k, vs in bw.items() if any(t in k for t in keys) for v in vs]
    if ff or ww:
        split[fam] = {"dispatches_per_step": len(ff) // 2, "fetch_bytes_per_step": sum(ff) * 1024 * cal["fetch_scalar"] / 2,
                      "write_bytes_per_step": sum(ww) * 1024 * cal["write_scalar"] / 2}
res = {"calibration_true_over_reported": cal, "conv_launches_per_step": launches, "per_family": split,
       "conv_fetch_bytes_per_step": fetch_bytes, "conv_write_bytes_per_step": write_bytes,
       "conv_hbm_bytes_per_launch": (fetch_bytes + write_bytes) / max(launches, 1),
       "note": "FETCH_SIZE/WRITE_SIZE (KiB) of all conv_igemm_kernel / conv_pipe_kernel / conv_smallmap_kernel / conv_wino(_ro|_rod|_rs)_kernel / wino4_input + wino4_gemm / conv_wino4f(_groups)_kernel / conv_bf16_kernel / conv_bf16_rv_kernel / conv_bf16_dg_kernel dispatches of one bench step, each corrected by the factor "
               "measured on fba_scalar_kernel (4-byte accesses, 1.07 GB known traffic); separate PMC passes"}
json.dump(res, open(f"{out}/conv_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
