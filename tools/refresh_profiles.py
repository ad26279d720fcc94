"""Copy the newest rocprofv3 outputs of tools/profile_round.sh from gpurun_out/prof6/ into profiles/ and rebuild the
derived JSON summaries (r1_traffic.json fp6 entry, r1_conv4_fp6_sq_summary.json).  profiles/README.md is edited by hand
around the tables this prints."""
import collections, csv, glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "gpurun_out", "prof6")
O = os.path.join(R, "profiles")


def newest(pattern):
    f = sorted(glob.glob(os.path.join(P, pattern)), key=os.path.getmtime)
    return f[-1]


def med(path, kern):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sorted(v)[len(v) // 2], len(v)) for k, v in acc.items()}


shutil.copy(newest("trace/runc/*_kernel_stats.csv"), os.path.join(O, "r1_bench_kernel_stats.csv"))
shutil.copy(newest("trace/runc/*_domain_stats.csv"), os.path.join(O, "r1_bench_domain_stats.csv"))
shutil.copy(os.path.join(P, "bench_under_prof.json"), os.path.join(O, "r1_bench_under_rocprof.json"))
for d, name in (("fetch", "r1_conv4_fp6_pmc_fetch_size.csv"), ("write", "r1_conv4_fp6_pmc_write_size.csv"),
                ("sq", "r1_conv4_fp6_pmc_sq.csv"), ("sq2", "r1_conv4_fp6_pmc_sq2.csv")):
    shutil.copy(newest(f"{d}/runc/*_counter_collection.csv"), os.path.join(O, name))

MAIN = "conv3x3_fp6_kernel"
corr = "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM) -> doubled; WRITE_SIZE exact"
t = json.load(open(os.path.join(O, "r1_traffic.json")))
f = med(os.path.join(O, "r1_conv4_fp6_pmc_fetch_size.csv"), MAIN)["FETCH_SIZE"]
w = med(os.path.join(O, "r1_conv4_fp6_pmc_write_size.csv"), MAIN)["WRITE_SIZE"]
lf = med(os.path.join(O, "r1_conv4_fp6_pmc_fetch_size.csv"), "lastpos")["FETCH_SIZE"]
lw = med(os.path.join(O, "r1_conv4_fp6_pmc_write_size.csv"), "lastpos")["WRITE_SIZE"]
key = [k for k in t if k.startswith("conv3x3_fp6_kernel")][0]
t[key].update({"FETCH_SIZE_KB_median": f[0], "WRITE_SIZE_KB_median": w[0], "launches": f[1],
               "lastpos_kernel_FETCH_SIZE_KB_median": lf[0], "lastpos_kernel_WRITE_SIZE_KB_median": lw[0],
               "hbm_bytes_per_launch_corrected": (2 * (f[0] + lf[0]) + w[0] + lw[0]) * 1024, "correction": corr})
json.dump(t, open(os.path.join(O, "r1_traffic.json"), "w"), indent=1)

sq = med(os.path.join(O, "r1_conv4_fp6_pmc_sq.csv"), MAIN); sq2 = med(os.path.join(O, "r1_conv4_fp6_pmc_sq2.csv"), MAIN)
d = {k: v[0] for k, v in {**sq, **sq2}.items()}
lp = {k: v[0] for k, v in med(os.path.join(O, "r1_conv4_fp6_pmc_sq.csv"), "lastpos").items()}
simds = 256 * 4
out = {"kernel": "conv3x3_fp6_kernel<6> (main kernel of den.conv4: Cout=512,Cin=256,B=256), medians over 13 launches; "
                 "the last-position kernel adds " + f"{lp.get('SQ_INSTS_VALU_MFMA_MOPS_F6F4', 0) * 512 / 1e12:.3f} TFLOP of MFMA work",
       "counters": d,
       "derived": {"kernel_cycles_per_XCD (GRBM_GUI_ACTIVE/8)": d["GRBM_GUI_ACTIVE"] / 8,
                   "mfma_busy_cycles_per_SIMD": d["SQ_VALU_MFMA_BUSY_CYCLES"] / simds,
                   "mfma_busy_fraction": d["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / (d["GRBM_GUI_ACTIVE"] / 8),
                   "executed_mfma_flops (MOPS_F6F4 x 512)": d["SQ_INSTS_VALU_MFMA_MOPS_F6F4"] * 512,
                   "dense_equivalent_flops": 473520144384.0,
                   "executed_over_dense_equivalent": (d["SQ_INSTS_VALU_MFMA_MOPS_F6F4"] + lp.get("SQ_INSTS_VALU_MFMA_MOPS_F6F4", 0)) * 512 / 473520144384.0,
                   "lds_array_busy_fraction_per_CU": d["SQ_LDS_IDX_ACTIVE"] / 256 / (d["GRBM_GUI_ACTIVE"] / 8),
                   "lds_bank_conflict_cycles": d["SQ_LDS_BANK_CONFLICT"]}}
json.dump(out, open(os.path.join(O, "r1_conv4_fp6_sq_summary.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1)); print(json.dumps(t[key], indent=1))
print(subprocess.run([sys.executable, os.path.join(R, "tools", "summarize_profile.py"), newest("trace/runc/*_kernel_trace.csv")],
                     capture_output=True, text=True).stdout)
