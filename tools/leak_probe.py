"""Does a handle pool leak?  Many short files, many batches, pools opened and closed: open file descriptors, threads, resident memory and
the device's free memory before and after.  usage: leak_probe.py [files] [batches]"""
import os, sys, time, tempfile, shutil, resource
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = tempfile.mkdtemp(prefix="jtleak", dir="/dev/shm")
e = Engine(0); paths = []
base = np.clip(np.rint(np.asarray(synth.speech_like(20.0, 48000, seed=3), np.float64) * 32768 * 2), -32768, 32767).astype(np.int16)
for k in range(NF):
    n = 48000 * 4 + (k * 7919) % (48000 * 12)
    p = os.path.join(d, f"s{k:04d}.flac"); open(p, "wb").write(e.op_flac_encode(base[:n], 48000, md5=True)); paths.append(p)
e.close()
def snap():
    st = open("/proc/self/status").read()
    thr = int([l for l in st.splitlines() if l.startswith("Threads:")][0].split()[1]); rss = int([l for l in st.splitlines() if l.startswith("VmRSS:")][0].split()[1]) // 1024
    free, total = torch.cuda.mem_get_info(0)
    return {"fds": len(os.listdir("/proc/self/fd")), "threads": thr, "rss_mb": rss, "gpu_used_mb": (total - free) >> 20}
print("start:", snap(), flush=True)
for rnd in range(3):
    with H.Pool((0,), 8) as P:
        for b in range(NB):
            t0 = time.time(); failed, fr, _ = P.process_files(paths, md5=True); w = time.time() - t0
            for r in fr:
                if r.rc == 0: os.unlink(r.output_path.decode())
        print(f"round {rnd}: {NB} batches of {NF} files, last {w * 1e3 / NF:.2f} ms/file, failed {failed}; inside the pool: {snap()}", flush=True)
    time.sleep(0.5)
    print(f"          after closing the pool: {snap()}", flush=True)
left = [q for q in os.listdir(d) if q.startswith(".processing-")]
print("residue:", left)
shutil.rmtree(d, ignore_errors=True)
