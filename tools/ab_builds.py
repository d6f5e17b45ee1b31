"""A/B of two or more BUILDS of the library (compile-time variants) on the bench file inside one gpurun call: each build runs in its own
process (JT_LIB_PATH), the builds alternate for `rounds` rounds so the box and its clocks are shared; step / pass / anlmdn / adeclick times.
usage: ab_builds.py [--rounds R] [--runs N] [--minutes M] lib_a.so lib_b.so ...        (child: ab_builds.py --child N M)"""
import os, sys, subprocess, json
HERE = os.path.dirname(os.path.abspath(__file__))


def child(n_runs, minutes):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.path.insert(0, os.path.join(HERE, ".."))
    import time
    import numpy as np, torch  # noqa: F401
    from jivetalking_amd import Engine, synth, hostlogic
    sr = 48000
    x = synth.speech_like_torch(minutes * 60.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
    e = Engine(0, ab=os.environ.get("JT_AB_CHILD") == "1")
    e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
    base = hostlogic.default_config()
    rows = []
    for i in range(n_runs + 2):
        t0 = time.perf_counter(); hostlogic.process_audio(e, base, 4096); dt = (time.perf_counter() - t0) * 1e3
        t = e.timers()
        if i >= 2: rows.append([dt, t["pass1_ms"], t["pass2_ms"], t["pass3_ms"], t["pass4_ms"], t["nlm_ms"], t["declick_ms"]])
    print("ABROWS " + json.dumps(rows))


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(int(a[1]), float(a[2])); sys.exit(0)
    rounds, runs, minutes = 3, 6, 60.0
    while a and a[0].startswith("--"):
        k, v = a[0], a[1]; a = a[2:]
        if k == "--rounds": rounds = int(v)
        elif k == "--runs": runs = int(v)
        elif k == "--minutes": minutes = float(v)
    import numpy as np
    res = {p: [] for p in a}
    for r in range(rounds):
        for p in a:
            # "lib.so@VAR=1,VAR2=x": environment of that variant's process (the A/B build imports JT_<KEY> at jt_open: "ab@JT_DK_SERIAL=1")
            path, _, assigns = p.partition("@")
            env = dict(os.environ)
            if path == "ab": env["JT_AB_CHILD"] = "1"
            else: env["JT_LIB_PATH"] = os.path.abspath(path)
            for a_ in filter(None, assigns.split(",")): k_, _, v_ = a_.partition("="); env[k_] = v_
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(runs), str(minutes)], env=env,
                                 capture_output=True, text=True)
            got = [ln for ln in out.stdout.splitlines() if ln.startswith("ABROWS ")]
            if not got:
                print(p, "FAILED:", out.stderr[-800:]); continue
            res[p] += json.loads(got[0][7:])
            if r == 0 and out.stderr.strip(): print(p, "stderr:", out.stderr.strip().splitlines()[-1][:600])
    for p in a:
        m = np.array(res[p])
        if not len(m): continue
        med, mn = np.median(m, axis=0), m.min(axis=0)
        print(f"{os.path.basename(p):28s} step {med[0]:6.2f} (min {mn[0]:6.2f})  pass 1/2/3/4 {med[1]:5.2f} / {med[2]:5.2f} / {med[3]:4.2f} / {med[4]:5.2f}"
              f"  anlmdn {med[5]:5.2f} (min {mn[5]:5.2f})  adeclick {med[6]:5.2f} (min {mn[6]:5.2f})  n={len(m)}")
