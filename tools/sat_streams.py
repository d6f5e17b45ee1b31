"""configs[3] on one GPU: N ten-minute FLAC files file to file through a handle pool, for every (streams per handle, files in flight).
usage: sat_streams.py [files] [md5 0/1] [streams list, e.g. 0,1,2] [in-flight list, e.g. 4,6,8,12]"""
import os, sys, time, tempfile, shutil, resource
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
from jivetalking_amd import _lib as L
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MD5 = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
SL = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2").split(",")]
KL = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "4,6,8,12").split(",")]
d = tempfile.mkdtemp(prefix="jtst", dir="/dev/shm")
if os.environ.get("JT_SCHED_FLAGS"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(int(os.environ["JT_SCHED_FLAGS"]))), flush=True)
e = Engine(0); paths = []
for k in range(NF):
    x = synth.speech_like_torch(600.0, 48000, seed=5000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
e.close()
torch.cuda.synchronize(); time.sleep(0.2)
lib = L.load()
import hashlib
ref_hash = None
for s in SL:
    for K in KL:
        lib.jt_set_option(None, b"pool_streams", str(s).encode())
        P = H.Pool((0,), K)
        P.process_files(paths[:K], md5=MD5)          # first-file allocations outside the timed batches
        best = []
        for rep in range(3):
            ru0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
            failed, fr, _ = P.process_files(paths, md5=MD5); w = time.time() - t0
            ru1 = resource.getrusage(resource.RUSAGE_SELF)
            cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
            best.append((w, cpu, failed))
        pstats = P.stats()
        P.close()
        hs = hashlib.md5()
        for r in fr: hs.update(open(r.output_path.decode(), "rb").read())
        if ref_hash is None: ref_hash = hs.hexdigest()
        same = hs.hexdigest() == ref_hash
        print(f"streams {s} in-flight {K:2d} md5 {int(MD5)} same_bytes {same}: " + "  ".join(f"{w / NF * 1e3:6.2f} ms/file (cpu {c / NF * 1e3:5.1f} ms/file, failed {f})" for w, c, f in best), flush=True)
        print("    per-file means of the last batch (ms):", pstats, flush=True)
shutil.rmtree(d, ignore_errors=True)
