"""jt_process_file on one 60-minute FLAC in /dev/shm: per-stage wall times (read, decode, encode, write).  usage: e2e_file.py [md5 0/1] [runs]"""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
md5 = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = tempfile.mkdtemp(prefix="jte2e", dir="/dev/shm")
x = synth.speech_like_torch(float(os.environ.get("JT_SECS", "3600")), 48000, seed=1000, device="cuda:0", plosives_per_min=40.0)
pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy(); del x
e = Engine(0)
if os.environ.get("JT_HT"): e.set_option("host_timing", "1")
for kv in filter(None, os.environ.get("JT_OPTS", "").split(",")):      # JT_OPTS=key=value,key=value
    e.set_option(*kv.split("=", 1))
p = os.path.join(d, "ep.flac"); open(p, "wb").write(e.op_flac_encode(pcm, 48000, md5=True))
for i in range(runs + 1):
    t0 = time.perf_counter(); res, out, io = H.process_file(e, p, md5=md5); dt = (time.perf_counter() - t0) * 1e3
    if i: print(f"file {dt:7.2f} ms   read {io[0]:.2f} decode {io[1]:.2f} encode {io[2]:.2f} write {io[3]:.2f}   passes+host {dt - sum(io):.2f}", flush=True)
    os.unlink(out)
shutil.rmtree(d, ignore_errors=True)
