"""ONE parametrised same-box A/B driver (replaces the per-experiment r4_run*.sh / r4_ab*.sh batches).

usage: python tools/ab.py [--passes 2] [--what layers,dense,elim,tests] <label>[:KEY=VAL[,KEY=VAL...]] ...

Every configuration is a label plus environment settings for a fresh child process: SPKDIFF_<OPTION>=<int> (forwarded to
spk_set_option by spkdiff/_lib.py: v2_duo, v2_waves, ...) and/or LIB=<path of a library variant built by tools/build_variant.sh>
(exported as SPKDIFF_LIB).  Per pass and configuration it runs
  layers: tools/fp6v2_time.py  -- per-layer time of spk_den_conv3x3_mfma_fp6v2 (main + tail launch) at B = 256 with the spike
          mismatch count against the six-plane kernel (bit-equality check),
  dense / elim: tools/listed_time.py -- the 100-step reverse process, dense (the bench line's mode) / elimination + lists,
  tests:  the fp6v2 bit-equality tests of the GPU suite.
Passes alternate over the configurations so that drift of the box shows.  Output: one block per (pass, label) on stdout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    passes, what = 2, ["layers", "dense"]
    while args and args[0].startswith("--"):
        if args[0] == "--passes":
            passes = int(args[1]); args = args[2:]
        elif args[0] == "--what":
            what = args[1].split(","); args = args[2:]
        else:
            raise SystemExit(__doc__)
    cfgs = []
    for spec in args:
        label, _, kv = spec.partition(":")
        env = {}
        for item in filter(None, kv.split(",")):
            k, _, v = item.partition("=")
            if k == "LIB":
                env["SPKDIFF_LIB"] = os.path.join(ROOT, v) if not os.path.isabs(v) else v
            else:
                env[k] = v
        cfgs.append((label, env))
    if not cfgs:
        raise SystemExit(__doc__)
    for ps in range(1, passes + 1):
        for label, env in cfgs:
            e = dict(os.environ, **env)
            print(f"== pass {ps} {label} {env}", flush=True)

            def run(cmd):
                r = subprocess.run(cmd, env=e, capture_output=True, text=True, cwd=ROOT)
                out = "\n".join(ln for ln in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in ln)
                print(out.strip()[-3000:], flush=True)
            if "layers" in what:
                run([sys.executable, os.path.join(ROOT, "tools", "fp6v2_time.py"), "--child"])
            if "dense" in what:
                run([sys.executable, os.path.join(ROOT, "tools", "listed_time.py"), "256", "3", "dense"])
            if "elim" in what:
                run([sys.executable, os.path.join(ROOT, "tools", "listed_time.py"), "256", "3", "elim+lists3"])
            if "tests" in what and ps == 1:
                run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x", "-k",
                     "fp6v2 or wide_dynamic or f5_denoiser or timed_configuration or skipping or split"])


if __name__ == "__main__":
    main()
