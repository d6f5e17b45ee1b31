"""One-GPU A/B of the pool's NUMA binding (option pool_numa, jt_handle_pool_open: worker + finisher threads and, through first touch,
the pinned I/O sets on the GPU's NUMA node): N ten-minute FLAC files file to file through a pool of eight one-stream handles, with the
STREAMINFO MD5 (the host-heavy variant), pool_numa off / on / off / on; ms per file, host CPU per file, bytes identical.
usage: ab_pool_numa.py [files]"""
import os, sys, time, tempfile, shutil, resource, hashlib, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
from jivetalking_amd import _lib as L
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = tempfile.mkdtemp(prefix="jtnuma", dir="/dev/shm")
e = Engine(0); paths = []
for k in range(NF):
    x = synth.speech_like_torch(600.0, 48000, seed=5000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
e.close()
torch.cuda.synchronize(); time.sleep(0.2)
lib = H.lib()
ncpu = C.c_int()
node = lib.jt_host_device_numa_node(C.c_int(0), C.byref(ncpu))
print(f"device 0: NUMA node {node}, {ncpu.value} of its CPUs allowed to this process; host threads {os.cpu_count()}", flush=True)
for f in sorted(os.listdir("/sys/devices/system/node")) if os.path.isdir("/sys/devices/system/node") else []:
    if f.startswith("node"):
        print("  ", f, open(f"/sys/devices/system/node/{f}/cpulist").read().strip(), flush=True)
ref = None
for rep in range(2):
    for numa in (0, 1):
        lib.jt_set_option(None, b"pool_numa", str(numa).encode())
        P = H.Pool((0,), 8)
        P.process_files(paths[:8], md5=True)
        runs = []
        for r in range(3):
            for q in os.listdir(d):
                if q.endswith("-processed.flac"): os.unlink(os.path.join(d, q))
            ru0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
            failed, fr, _ = P.process_files(paths, md5=True); w = time.time() - t0
            ru1 = resource.getrusage(resource.RUSAGE_SELF)
            runs.append((w / NF * 1e3, ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / NF * 1e3, failed))
        st = P.stats(); P.close()
        hs = hashlib.md5()
        for r in fr: hs.update(open(r.output_path.decode(), "rb").read())
        ref = ref or hs.hexdigest()
        print(f"pool_numa {numa}: " + "  ".join(f"{w:6.2f} ms/file (cpu {c:5.1f} ms/file, failed {f})" for w, c, f in runs) + f"  same bytes {hs.hexdigest() == ref}", flush=True)
        print("    per-file means of the last batch (ms):", st, flush=True)
shutil.rmtree(d, ignore_errors=True)
