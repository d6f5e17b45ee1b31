"""A handle pool fed a random mix: mono and stereo FLAC, 16 / 24 bit, WAV, lengths from 1 s to 12 minutes, files that take the dynamic
mode, silent files, truncated and corrupted files, a file that is not audio.  Every file's outcome -- bytes of the output, or the error
code -- must be what jt_process_file gives for that file alone on a fresh handle, in every batch, for several pool sizes.
usage: fuzz_pool.py [files] [seed]"""
import os, sys, time, tempfile, shutil, hashlib, struct
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
from oracle import orc
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
d = tempfile.mkdtemp(prefix="jtfz", dir="/dev/shm")
e = Engine(0); paths = []
def wav(x16, rate, ch):
    payload = x16.astype("<i2").tobytes(); align = 2 * ch
    fmt = struct.pack("<HHIIHH", 1, ch, rate, rate * align, align, 16)
    body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body
for k in range(NF):
    kind = int(rng.integers(0, 10)); secs = float(np.exp(rng.uniform(np.log(1.0), np.log(720.0)))); rate = int(rng.choice([48000, 48000, 44100]))
    x = synth.speech_like_torch(secs, rate, seed=int(rng.integers(1, 10**6)), device="cuda:0", plosives_per_min=float(rng.choice([0.0, 40.0])),
                                sib_gain=4.0 if kind == 1 else 0.25).cpu().numpy().astype(np.float64) * float(10 ** rng.uniform(-0.7, 0.2))
    x16 = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
    name = os.path.join(d, f"f{k:03d}")
    if kind == 2:                                            # stereo 16-bit FLAC from the oracle's coverage encoder
        st = np.stack([x16, (x16 * 0.6).astype(np.int16)], axis=1).astype(np.int32)[: 20 * rate]
        data, name = orc.flac_encode(st, rate, 16, 4096, 2 | 64, 8), name + ".flac"
    elif kind == 3:                                          # 24-bit mono FLAC
        x24 = (x16.astype(np.int32) << 8)[: 20 * rate, None]
        data, name = orc.flac_encode(x24, rate, 24, 4096, 2, 8), name + ".flac"
    elif kind == 4: data, name = wav(x16, rate, 1), name + ".wav"
    elif kind == 5: data, name = e.op_flac_encode(np.zeros(int(rate * min(secs, 30)), np.int16), rate, md5=True), name + ".flac"      # silence
    elif kind == 6:
        data = bytearray(e.op_flac_encode(x16, rate, md5=True)); data[len(data) // 2] ^= 0x20; data, name = bytes(data), name + ".flac"
    elif kind == 7: data = e.op_flac_encode(x16, rate, md5=True); data, name = data[: len(data) * 2 // 3], name + ".flac"            # truncated
    elif kind == 8 and k % 3 == 0: data, name = b"this is not audio" * 50, name + ".flac"
    else: data, name = e.op_flac_encode(x16, rate, md5=True), name + ".flac"
    open(name, "wb").write(data); paths.append(name)
# what each file gives alone
want = []
for p in paths:
    try:
        res, out, _ = H.process_file(e, p, md5=True)
        want.append((0, hashlib.md5(open(out, "rb").read()).hexdigest())); os.unlink(out)
    except L.JtError as ex:
        want.append((ex.code, None))
e.close()
print("alone:", sum(1 for w in want if w[0] == 0), "succeed,", sum(1 for w in want if w[0] != 0), "fail with", sorted({w[0] for w in want if w[0] != 0}), flush=True)
bad = 0
for K in (3, 8, 5):
    with H.Pool((0,), K) as P:
        for b in range(3):
            order = rng.permutation(NF)
            failed, fr, _ = P.process_files([paths[i] for i in order], md5=True)
            for j, i in enumerate(order):
                r = fr[j]
                got = (0, hashlib.md5(open(r.output_path.decode(), "rb").read()).hexdigest()) if r.rc == 0 else (r.rc, None)
                if r.rc == 0: os.unlink(r.output_path.decode())
                if got != want[i]:
                    bad += 1; print(f"  pool of {K}, batch {b}: {os.path.basename(paths[i])} gives {got}, alone {want[i]}; error text {r.error!r}", flush=True)
            print(f"pool of {K}, batch {b}: {failed} failed of {NF}", flush=True)
left = [q for q in os.listdir(d) if q.startswith(".processing-")]
print("residue:", left, " RESULT:", "FAILED" if bad or left else "ok")
shutil.rmtree(d, ignore_errors=True)
