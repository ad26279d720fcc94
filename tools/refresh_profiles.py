"""usage: python tools/refresh_profiles.py <tag>   (e.g. r5)
Copy the newest rocprofv3 outputs of tools/profile_pass.sh <tag> from gpurun_out/<tag>prof/ into profiles/ (<tag>_ names) and
rebuild the derived JSON summaries: <tag>_traffic.json (per-launch HBM-side bytes, corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE doubled, WRITE_SIZE as is; KB = 1024 B; each entry carries the sha256 of the kernel source it was measured on and
the commit) and <tag>_conv4_fp6v2_sq_summary.json (matrix-pipe / LDS utilisation)."""
import collections, csv, glob, hashlib, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
P = os.path.join(R, "gpurun_out", TAG + "prof")
O = os.path.join(R, "profiles")


def newest(pattern):
    return sorted(glob.glob(os.path.join(P, pattern)), key=os.path.getmtime)[-1]


def med(path, kern):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sorted(v)[len(v) // 2], len(v)) for k, v in acc.items()}


def sha16(rel):
    return hashlib.sha256(open(os.path.join(R, "spiking-diffusion_amd", "csrc", rel), "rb").read()).hexdigest()[:16]


commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
shutil.copy(newest("trace/runc/*_kernel_stats.csv"), os.path.join(O, TAG + "_bench_kernel_stats.csv"))
shutil.copy(newest("trace/runc/*_domain_stats.csv"), os.path.join(O, TAG + "_bench_domain_stats.csv"))
shutil.copy(os.path.join(P, "bench_under_prof.json"), os.path.join(O, TAG + "_bench_under_rocprof.json"))
for src, dst in (("trace_headline/runc/*_kernel_stats.csv", TAG + "_bench_headline_kernel_stats.csv"),
                 ("trace_encdec/runc/*_kernel_stats.csv", TAG + "_encdec_kernel_stats.csv"),
                 ("trace_lists/runc/*_kernel_stats.csv", TAG + "_lists_kernel_stats.csv"),
                 ("trace_mainpy/runc/*_kernel_stats.csv", TAG + "_main_py_shape_kernel_stats.csv")):
    try:
        shutil.copy(newest(src), os.path.join(O, dst))
    except IndexError:
        pass
for src, dst in (("bench_headline_under_prof.json", TAG + "_bench_headline_under_rocprof.json"),
                 ("bench_encdec_under_prof.json", TAG + "_encdec_under_rocprof.json"), ("listed_time.log", TAG + "_lists_listed_time.log"),
                 ("bench_mainpy_under_prof.json", TAG + "_main_py_shape_under_rocprof.json")):
    if os.path.exists(os.path.join(P, src)):
        shutil.copy(os.path.join(P, src), os.path.join(O, dst))
names = {"v2_FETCH_SIZE": TAG + "_conv4_fp6v2_pmc_fetch_size.csv", "v2_WRITE_SIZE": TAG + "_conv4_fp6v2_pmc_write_size.csv",
         "v2c5_FETCH_SIZE": TAG + "_conv5_fp6v2_pmc_fetch_size.csv", "v2c5_WRITE_SIZE": TAG + "_conv5_fp6v2_pmc_write_size.csv",
         "v2cifar_FETCH_SIZE": TAG + "_conv4_8x8_fp6v2_pmc_fetch_size.csv", "v2cifar_WRITE_SIZE": TAG + "_conv4_8x8_fp6v2_pmc_write_size.csv",
         "v2_sq": TAG + "_conv4_fp6v2_pmc_sq.csv", "v2_sq2": TAG + "_conv4_fp6v2_pmc_sq2.csv",
         "encdec_FETCH_SIZE": TAG + "_encdec_pmc_fetch_size.csv", "encdec_WRITE_SIZE": TAG + "_encdec_pmc_write_size.csv",
         "lif_FETCH_SIZE": TAG + "_lif_pmc_fetch_size.csv", "lif_WRITE_SIZE": TAG + "_lif_pmc_write_size.csv",
         "encdec_sq": TAG + "_encdec_pmc_sq.csv", "tail_FETCH_SIZE": TAG + "_step_tail_pmc_fetch_size.csv",
         "tail_WRITE_SIZE": TAG + "_step_tail_pmc_write_size.csv", "tail_sq": TAG + "_step_tail_pmc_sq.csv"}
for d, name in names.items():
    shutil.copy(newest(f"{d}/runc/*_counter_collection.csv"), os.path.join(O, name))

corr = "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM) -> doubled; WRITE_SIZE exact"
t = {}


def entry(key, fetch_csv, write_csv, kernels, src, extra=None):
    f = [med(os.path.join(O, fetch_csv), k).get("FETCH_SIZE", (0.0, 0)) for k in kernels]
    w = [med(os.path.join(O, write_csv), k).get("WRITE_SIZE", (0.0, 0)) for k in kernels]
    e = {"kernels": kernels, "FETCH_SIZE_KB_median": [x[0] for x in f], "WRITE_SIZE_KB_median": [x[0] for x in w],
         "launches": f[0][1], "hbm_bytes_per_launch_corrected": (2 * sum(x[0] for x in f) + sum(x[0] for x in w)) * 1024,
         "correction": corr, "kernel_source": src, "kernel_source_sha16": sha16(src), "commit": commit}
    e.update(extra or {})
    t[key] = e


entry("conv3x3_fp6v2_kernel + tail launches (Cout=512,Cin=256,B=256: den.conv4 shape)", TAG + "_conv4_fp6v2_pmc_fetch_size.csv",
      TAG + "_conv4_fp6v2_pmc_write_size.csv", ["conv3x3_fp6v2_kernel", "fp6v2_tail_kernel"],
      "den_mfma_fp6v2.hip",
      {"algorithmic_bytes_per_launch": 256 * 49 * 16 * (256 // 2 + 512 // 2) + 16 * 8 * 38912,
       "note": "algorithmic = input spikes (fp4, S32) + output spikes once + packed weights once"})
entry("conv3x3_fp6v2_kernel + tail launches (Cout=256,Cin=512,B=256: den.conv5 shape)", TAG + "_conv5_fp6v2_pmc_fetch_size.csv",
      TAG + "_conv5_fp6v2_pmc_write_size.csv", ["conv3x3_fp6v2_kernel", "fp6v2_tail_kernel"], "den_mfma_fp6v2.hip",
      {"algorithmic_bytes_per_launch": 256 * 49 * 16 * (512 // 2 + 256 // 2) + 256 * 49 * 256 + 8 * 16 * 38912,
       "note": "algorithmic = input spikes (fp4, S32) + output spikes and spike counts once + packed weights once"})
entry("conv3x3_fp6v2_kernel<8, 8> + repair launch (Cout=512,Cin=256,B=512, 8x8 latents: configs[3] den.conv4 shape)",
      TAG + "_conv4_8x8_fp6v2_pmc_fetch_size.csv", TAG + "_conv4_8x8_fp6v2_pmc_write_size.csv", ["conv3x3_fp6v2_kernel", "fp6v2_tail_kernel"],
      "den_mfma_fp6v2.hip", {"algorithmic_bytes_per_launch": 512 * 64 * 16 * (256 // 2 + 512 // 2) + 16 * 8 * 38912,
                             "note": "algorithmic = input spikes (fp4, S32) + output spikes once + packed weights once"})
entry("vae.dec2: vae_fp6_kernel<0, 14, 14, 2, 0, 2, false> + repair launch (decoder convT2 64->32, 14x14 -> 28x28, B=1024)",
      TAG + "_encdec_pmc_fetch_size.csv", TAG + "_encdec_pmc_write_size.csv", ["vae_fp6_kernel<0, 14, 14, 2", "vae_fp6_fixup_kernel<0, 14, 14, 2"],
      "vae_fp6.hip", {"algorithmic_bytes_per_launch": 1024 * (196 * 16 * 32 + 784 * 32 * 4) + 45 * 1536,
                      "note": "algorithmic = S32 input spikes (1/2 B per neuron-step) + fp32 time-collapsed output + packed weights once"})
entry("lif_fwd_kernel (T=16,N=25.7M)", TAG + "_lif_pmc_fetch_size.csv", TAG + "_lif_pmc_write_size.csv", ["lif_fwd_kernel"], "lif.hip",
      {"algorithmic_bytes_per_launch": 8 * 16 * 1024 * 32 * 28 * 28 + 8 * 1024 * 32 * 28 * 28})
vq = med(os.path.join(O, TAG + "_encdec_pmc_sq.csv"), "vae_fp6_kernel<0, 14, 14, 2")
for k in t:
    if k.startswith("vae.dec2"):
        t[k]["valu_insts_per_launch"] = vq.get("SQ_INSTS_VALU", (0.0, 0))[0]
        t[k]["sq_counters_median"] = {c: v[0] for c, v in vq.items()}
        ns = 1024 * 784 * 32 * 16
        t[k]["valu_insts_per_neuron_step"] = vq.get("SQ_INSTS_VALU", (0.0, 0))[0] * 64.0 / ns
entry("step_tail_kernel<7, 7> (conv6 on counts + token update + next conv1, B=256)", TAG + "_step_tail_pmc_fetch_size.csv",
      TAG + "_step_tail_pmc_write_size.csv", ["step_tail_kernel<7, 7"], "step_tail.hip",
      {"algorithmic_bytes_per_launch": 256 * 49 * 32 * 10 + 8 * 10 * 18432 + 256 * 49 * (16 * 16 * 2 + 64) + 256 * 49 * 9,
       "note": "algorithmic = count records in + packed conv6 weights once + next conv1 spikes and counts out + token state",
       "sq_counters_median": {c: v[0] for c, v in med(os.path.join(O, TAG + "_step_tail_pmc_sq.csv"), "step_tail_kernel<7, 7").items()}})
json.dump(t, open(os.path.join(O, TAG + "_traffic.json"), "w"), indent=1)

MAIN = "conv3x3_fp6v2_kernel"
sq = med(os.path.join(O, TAG + "_conv4_fp6v2_pmc_sq.csv"), MAIN); sq2 = med(os.path.join(O, TAG + "_conv4_fp6v2_pmc_sq2.csv"), MAIN)
d = {k: v[0] for k, v in {**sq, **sq2}.items()}
simds = 256 * 4
mfma_per_simd = 16 * 8 * 18 * 6                   # items x chunks x MFMAs per tile and chunk x tiles per SIMD (two waves of three)
out = {"kernel": "conv3x3_fp6v2_kernel<7,7,8> (main launch of den.conv4: Cout=512,Cin=256,B=256; four digits, two waves per SIMD), medians over the launches",
       "counters": d,
       "derived": {"kernel_cycles_per_XCD (GRBM_GUI_ACTIVE/8)": d["GRBM_GUI_ACTIVE"] / 8,
                   "mfma_busy_cycles_per_SIMD": d["SQ_VALU_MFMA_BUSY_CYCLES"] / simds,
                   "mfma_busy_fraction": d["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / (d["GRBM_GUI_ACTIVE"] / 8),
                   "mfma_instructions_per_SIMD (expected %d)" % mfma_per_simd: d.get("SQ_INSTS_MFMA", 0) / simds,
                   "executed_mfma_flops (MOPS_F6F4 x 512)": d["SQ_INSTS_VALU_MFMA_MOPS_F6F4"] * 512,
                   "dense_equivalent_flops": 473520144384.0,
                   "executed_over_dense_equivalent": d["SQ_INSTS_VALU_MFMA_MOPS_F6F4"] * 512 / 473520144384.0,
                   "mfma_coexec_fraction_of_busy": d.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(d["SQ_VALU_MFMA_BUSY_CYCLES"], 1),
                   "lds_array_busy_fraction_per_CU": d["SQ_LDS_IDX_ACTIVE"] / 256 / (d["GRBM_GUI_ACTIVE"] / 8),
                   "lds_bank_conflict_cycles": d["SQ_LDS_BANK_CONFLICT"]},
       "kernel_source_sha16": sha16("den_mfma_fp6v2.hip"), "commit": commit}
json.dump(out, open(os.path.join(O, TAG + "_conv4_fp6v2_sq_summary.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
for k, v in t.items():
    print(k, "->", round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1), "MB per launch; algorithmic",
          round(v.get("algorithmic_bytes_per_launch", 0) / 1e6, 1), "MB")
shutil.copy(newest("trace_headline/runc/*_kernel_stats.csv"), os.path.join(O, TAG + "_bench_headline_kernel_stats.csv"))
shutil.copy(os.path.join(P, "bench_headline_under_prof.json"), os.path.join(O, TAG + "_bench_headline_under_rocprof.json"))
per_layer = subprocess.run([sys.executable, os.path.join(R, "tools", "summarize_profile.py"),
                            newest("trace_headline/runc/*_kernel_trace.csv")], capture_output=True, text=True).stdout
open(os.path.join(O, TAG + "_headline_per_layer.md"), "w").write(per_layer)
print(per_layer)
