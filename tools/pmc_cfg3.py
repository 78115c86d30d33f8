"""Per-kernel HBM traffic of a cfg3 step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), keyed by the library's
launch names (gemm_nn / gemm_nt / gemm_tn / smpf_* ...), per step.  usage: pmc_cfg3.py FETCH_DIR WRITE_DIR STEPS OUT.json
Corrections per MI355X_MICROARCH.md (HBM / rocprofv3): counters in KiB; FETCH_SIZE x2 on gfx950, WRITE_SIZE x1."""
import csv, glob, json, re, sys, collections


def launch_name(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gf::gemm_f32_mfma(_grouped)?<(true|false), (true|false)", k)
    if m:
        return "gemm_" + ("t" if m.group(2) == "true" else "n") + ("t" if m.group(3) == "true" else "n")
    m = re.match(r"gf::gemm_f32_mfma_free<(true|false), (true|false)", k)
    if m:
        return "smpf_small_" + ("t" if m.group(1) == "true" else "n") + ("t" if m.group(2) == "true" else "n")
    m = re.match(r"gf::smp_rowpanel_(?:c64|split)<(true|false)", k)
    if m:
        return "smpf_products_fwd" if m.group(1) == "true" else "smpf_products_bwd"
    m = re.match(r"gf::fam50_bwd_tables_mfma<\d+, (\d)", k)
    if m:
        return "fam_bwd_tables_col" if m.group(1) == "0" else "fam_bwd_tables_row"
    base = re.sub(r"[<(].*", "", k).split("::")[-1]
    table = {"smp_tables_fwd": "smpf_tables_fwd", "smp_tables_fwd_w": "smpf_tables_fwd", "smp_tables_bwd": "smpf_tables_bwd",
             "smp_combine_fwd": "smpf_combine_fwd", "smp_combine_bwd": "smpf_combine_bwd", "smp_combine_fwd_panels": "smpf_combine_fwd",
             "smp_combine_bwd_panels": "smpf_combine_bwd", "smp_bwd_gather_all": "smpf_bwd_gather", "smp_small_split": "smpf_small_split", "promote_backward": "smp_promote_bwd",
             "smp_vectors": "smpf_vectors", "smp_bwd_gather": "smpf_bwd_gather", "smp_wgrad_c64": "smpf_wgrad", "smp_wgrad_split": "smpf_wgrad",
             "smp_reduce_pairs": "smpf_reduce_pairs", "smp_fold_level": "smpf_fold", "diag_gather_bwd": "smpf_diag_gather_bwd",
             "diag_gather_fwd": "smpf_diag_gather", "stack_weights_all": "smpf_stack_w", "readout_nodes_v": "smp_readout_nodes",
             "fam_backward_rows": "fam_backward", "fam_backward_cols": "fam_backward", "fam_products_lds": "fam_products",
             "fam50_tables_out": "fam_tables", "fam50_forward_mfma": "fam_forward", "fam50_bwd_scalars_fold": "fam_bwd_scalars"}
    return table.get(base, base)


def collect(d, ctr, scale):
    tot = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                tot[launch_name(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024 * scale
    return tot


def is_setup(name):
    """Kernels of the process's set-up (input uploads, gf_smp_prepare's table builders, the bench's own copies): they run once per
    prepared batch, not once per step, and stay out of the per-step totals (round-5 review, weak #12)."""
    return name.startswith(("__amd_rocclr_", "build_", "invert_cons", "level_table_stats", "tables_zero_fill", "copy_probe", "reduce_kernel", "triu_tril")) or "elementwise" in name or "distribution" in name


fd, wd, steps, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
commit = sys.argv[5] if len(sys.argv) > 5 else None
fetch, write = collect(fd, "FETCH_SIZE", 2.0), collect(wd, "WRITE_SIZE", 1.0)
names = sorted(set(fetch) | set(write))
res = {k: {"fetch": fetch.get(k, 0) / steps, "write": write.get(k, 0) / steps} for k in names if not is_setup(k)}
setup = {k: {"fetch": fetch.get(k, 0), "write": write.get(k, 0)} for k in names if is_setup(k)}
res["_setup_per_process"] = setup
res["_meta"] = {"commit": commit, "steps_per_process": steps, "units": "bytes per step (FETCH_SIZE x 2, WRITE_SIZE x 1, KiB counters); _setup_per_process: bytes per process"}
json.dump(res, open(out, "w"), indent=1)
step = {k: v for k, v in res.items() if not k.startswith("_")}
tot = sum(v["fetch"] + v["write"] for v in step.values())
for k, v in sorted(step.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"]))[:16]:
    print(f"{k:22s} read {v['fetch']/1e9:7.3f} GB  write {v['write']/1e9:7.3f} GB  per step")
print(f"total {tot/1e9:.2f} GB per step (set-up kernels excluded: {sum(v['fetch'] + v['write'] for v in setup.values())/1e9:.2f} GB per process)")
if commit:
    print("commit", commit)
