"""k_finish ablation through the variants of the DIAGNOSIS build (`make -C vfind_amd/csrc ablation` ->
vfind_amd/libvfind_hip_ablation.so; the product library does not contain them), chosen by VFIND_FINISH_DBG:
0 = product path, 1 = no slot-line loads, 2 = no row loads, 3 = no slot-line loads and no counting,
4 = slot-line loads but no counting.  Modes 1-4 give wrong counts by construction (profiling only).
Each mode runs bench.py in its own process, cold (table cleared every step) and warm (table kept).
Round 1 on one MI355X, cfg 2: warm 0.547 / 0.518 / 0.646 / 0.063 / 0.271 ms -> counting, not probing,
was the cost; that is what moved the counts to per-XCD arrays with L2-local atomics."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for keep in ("0", "1"):
    for dbg in ("0", "1", "2", "3", "4"):
        env = dict(os.environ, VFIND_FINISH_DBG=dbg, VFIND_BENCH_KEEP_TABLE=keep, VFIND_RES_LANES="1")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-parity",
                              "--lib", os.path.join(root, "vfind_amd", "libvfind_hip_ablation.so")], env=env,
                             capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(f"table {'kept' if keep == '1' else 'cleared'} dbg={dbg}: finish {d['roofline_path']['kernel_ms_per_step']['finish']:.3f} ms, step {d['ms_per_step']:.3f} ms", flush=True)
