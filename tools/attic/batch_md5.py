"""jt_process_files on copies of one FLAC (JT_BATCH_MINUTES, default 60; JT_BATCH_MD5=0 switches the MD5 off) in /dev/shm, MD5 on: ms per file against files in flight (the MD5 of a file is one
host core for ~380 ms; workers hide it behind each other's GPU phases).  python tools/batch_md5.py [files] [in_flight,...]"""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jivetalking_amd import Engine, synth, hostlogic
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3,6,8").split(",")]
sr = 48000
mins = float(os.environ.get("JT_BATCH_MINUTES", "60"))
md5 = os.environ.get("JT_BATCH_MD5", "1") == "1"
x = synth.speech_like_torch(mins * 60.0, sr, seed=1000, device="cuda:0")
eng = Engine(0)
pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
d = tempfile.mkdtemp(prefix="jtbatch", dir="/dev/shm")
try:
    src = os.path.join(d, "episode.flac"); open(src, "wb").write(eng.op_flac_encode(pcm, sr, md5=True))
    base = hostlogic.default_config()
    for k in ks:
        for rep in range(2):                     # the second round of a setting reuses nothing: handles are per call
            paths = []
            for i in range(nb):
                pk = os.path.join(d, f"b{i}.flac"); shutil.copyfile(src, pk); paths.append(pk)
            t0 = time.perf_counter()
            failed, fr = hostlogic.process_files(paths, device=0, in_flight=k, base=base, md5=md5)
            tb = time.perf_counter() - t0
            print(f"in_flight {k}: {nb} files, failed {failed}, {tb / nb * 1e3:.1f} ms per file ({nb * mins * 60.0 / tb:.0f} xRT)", flush=True)
            for f in os.listdir(d):
                if f != "episode.flac": os.remove(os.path.join(d, f))
finally:
    shutil.rmtree(d, ignore_errors=True)
