"""Soak: the same 24 ten-minute FLAC files through jt_process_files_multi again and again (8 in flight, MD5 on); every round's output
files must be byte-identical to the first round's and every measurement equal.  usage: soak_batch.py [rounds]"""
import os, sys, time, tempfile, shutil, hashlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = tempfile.mkdtemp(prefix="jtsoak", dir="/dev/shm")
e = Engine(0); paths = []
nf = 24
for k in range(nf):
    x = synth.speech_like_torch(600.0, 48000, seed=7000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
e.close()
ref = None; bad = 0; t_all = time.time()
for r in range(rounds):
    t0 = time.time(); failed, fr, _ = H.process_files_multi(paths, devices=(0,), in_flight_per_device=8, md5=True); w = time.time() - t0
    got = []
    for i in range(nf):
        out = fr[i].output_path.decode()
        got.append((hashlib.md5(open(out, "rb").read()).hexdigest(), fr[i].result.output_lufs, fr[i].result.output_tp_db, fr[i].result.measure.input_i,
                    fr[i].result.final_speech.rms_level, fr[i].result.filtered_room_tone.rms_level, int(fr[i].result.limiter.needed)))
        os.unlink(out)
    if ref is None: ref = got
    diff = sum(1 for a, b in zip(ref, got) if a != b)
    bad += diff + failed
    print(f"round {r}: {w / nf * 1e3:.1f} ms per file, failed {failed}, differing from round 0: {diff}", flush=True)
print(f"{rounds * nf} files in {time.time() - t_all:.1f} s, bad {bad}")
shutil.rmtree(d, ignore_errors=True)
sys.exit(1 if bad else 0)
