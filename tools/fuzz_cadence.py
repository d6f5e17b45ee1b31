"""Round-6 code on random material: files of random container / sample format / rate / channel count and layout / block size (incl.
variable-blocksize FLAC) / length through jt_load_audio + jt_analyse_only with frame_samples = 0.  Checked per file: the cadence the
library reports against the rule (FLAC: the stream's frames; WAV: 4096-byte packets of whole sample blocks), the interval series
(count, timestamps, per-interval RMS) against analyser.go:588-600 restated here, the down-mix (astats Min / Max level bit-exact against the
oracle's matrix), and -- every fourth file -- the whole job twice (identical bytes) with the landing re-measured by the oracle.
usage: fuzz_cadence.py [cases] [seed]"""
import sys, struct, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
from oracle import orc
MASKS = {1: [0x4], 2: [0x3], 3: [0x7, 0xB, 0], 4: [0x107, 0x33, 0x603, 0], 5: [0x607, 0x37, 0], 6: [0x60F, 0x3F, 0], 7: [0x70F, 0], 8: [0x63F, 0xFF, 0]}
FLAC_LAYOUT = [0, 0x4, 0x3, 0x7, 0x33, 0x607, 0x60F, 0x70F, 0x63F]


def starts(frame_lens, sr):
    out = []; start = 0; processed = 0; acc = 0
    for nb in frame_lens:
        t = int(float(processed) / float(sr) * 1e9)
        processed += int(nb); acc += int(nb)
        if t - start >= 250_000_000:
            out.append((start, acc)); start = t; acc = 0
    if acc > 0:
        out.append((start, acc))
    return out


def wav(x, rate, ch, kind, mask):
    if kind == "f32": payload, tag, bits = np.asarray(x, "<f4").tobytes(), 3, 32
    elif kind == "f64": payload, tag, bits = np.asarray(x, "<f8").tobytes(), 3, 64
    elif kind == "s16": payload, tag, bits = np.clip(np.rint(np.asarray(x, np.float64) * 32768), -32768, 32767).astype("<i2").tobytes(), 1, 16
    elif kind == "u8": payload, tag, bits = (np.clip(np.rint(np.asarray(x, np.float64) * 128), -128, 127) + 128).astype(np.uint8).tobytes(), 1, 8
    else:
        v = np.clip(np.rint(np.asarray(x, np.float64) * 8388608), -8388608, 8388607).astype("<i4")
        payload, tag, bits = v.view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 1, 24
    align = ch * bits // 8
    if mask:
        fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, ch, rate, rate * align, align, bits, 22, bits, mask, tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * align, align, bits)
    body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body, align


def run(e, cases, seed, verbose=True):
  rng = np.random.default_rng(seed)
  bad = 0; failures = []
  for c in range(cases):
      sr = int(rng.choice([44100, 48000, 96000, 32000, 22050]))
      ch = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 7, 8]))
      secs = float(rng.uniform(3.0, 40.0)) * (0.5 if sr == 96000 else 1.0)
      n = int(secs * sr)
      base = np.asarray(synth.speech_like(n / sr + 0.3, sr, seed=int(rng.integers(1, 10**6))), np.float64)
      x = np.empty((n, ch), np.float64)
      for k in range(ch):
          x[:, k] = np.roll(base, 41 * k)[:n] * (0.9 - 0.07 * k) + rng.standard_normal(n) * 0.002
      x = np.clip(x, -0.999, 0.999)
      issues = []
      if rng.random() < 0.5:                                     # FLAC
          bps = int(rng.choice([16, 16, 24]))
          bs = int(rng.choice([576, 1152, 2304, 4096, 4608, 1024, 192]))
          variable = rng.random() < 0.3
          q = float(1 << (bps - 1))
          pcm = np.clip(np.rint(x * q), -q, q - 1).astype(np.int32)
          data = orc.flac_encode(pcm, sr, bps, bs, 2 | (32 if variable else 0), 8)
          raw = (pcm.astype(np.float64) / q).astype(np.float32).reshape(-1)
          mask = FLAC_LAYOUT[ch]
          if variable:
              minbs = max(bs // 2, 16); lens = []; done = 0; k = 0
              while done < n:
                  b = min(minbs if (k & 1) else bs, n - done); lens.append(b); done += b; k += 1
          else:
              lens = [bs] * (n // bs) + ([n % bs] if n % bs else [])
          what = f"flac {bps} bit bs {bs}{' variable' if variable else ''}"
      else:
          kind = str(rng.choice(["s16", "f32", "s24", "u8", "f64"]))
          mask = int(rng.choice(MASKS[ch]))
          data, align = wav(x.reshape(-1), sr, ch, kind, mask)
          per = max(align, 4096) // align if align > 1 else 4096
          lens = [per] * (n // per) + ([n % per] if n % per else [])
          if kind == "f32": raw = x.reshape(-1).astype(np.float32)
          elif kind == "f64": raw = x.reshape(-1).astype(np.float32)
          elif kind == "s16": raw = (np.clip(np.rint(x.reshape(-1) * 32768), -32768, 32767) / 32768.0).astype(np.float32)
          elif kind == "u8": raw = (np.clip(np.rint(x.reshape(-1) * 128), -128, 127) / 128.0).astype(np.float32)
          else: raw = (np.clip(np.rint(x.reshape(-1) * 8388608), -8388608, 8388607) / 8388608.0).astype(np.float32)
          what = f"wav {kind} mask {mask:#x}"
      try:
          meta = e.load_audio(data)
      except L.JtError as ex:
          failures.append((c, what, sr, ch, [f"load failed {ex}"])); bad += 1; continue
      fs, var, nfr, glens = e.input_frame_layout()
      got_lens = glens.tolist() if var else [fs] * (n // fs) + ([n % fs] if n % fs else [])
      if got_lens != lens: issues.append(f"cadence {fs} var {var} frames {nfr} vs {len(lens)}")
      if meta["channel_mask"] != (mask or int(orc.lib().orc_default_layout(C.c_int(ch)) if False else [0, 0x4, 0x3, 0xB, 0x107, 0x37, 0x3F, 0x70F, 0x63F][ch])): issues.append(f"mask {meta['channel_mask']:#x}")
      g = H.process_audio(e, frame_samples=0, analyse_only=True)
      iv = (H.Interval * 4096)(); niv = H.lib().jt_host_last_intervals(e.h, iv, C.c_int64(4096))
      want = starts(lens, sr)
      if niv != len(want) or [iv[i].timestamp_ns for i in range(niv)] != [w[0] for w in want]: issues.append("interval series")
      else:
          r64 = raw.astype(np.float64); pos = 0
          for i, (_, cnt) in enumerate(want):
              seg = r64[pos * ch:(pos + cnt) * ch]; pos += cnt
              rms = float(np.sqrt(np.mean(seg * seg))); ref = -120.0 if rms < 1e-5 else 20 * np.log10(rms)
              if abs(iv[i].rms_level - ref) > 1e-9: issues.append(f"interval {i} rms"); break
      mono = orc.downmix_layout(raw, ch, mask, 0) if ch > 1 else raw
      p1 = e.pass1(n, sample_rate=sr)
      if (p1["astats"]["max_level"], p1["astats"]["min_level"]) != (float(mono.max()), float(mono.min())):
          issues.append(f"down-mix extremes {p1['astats']['max_level']} {p1['astats']['min_level']} vs {float(mono.max())} {float(mono.min())}")
      if c % 4 == 0 and sr >= 44100 and secs >= 12:
          try:
              r1 = H.process_audio(e, frame_samples=0); o1 = e.download_s16(4).copy()
              r2 = H.process_audio(e, frame_samples=0); o2 = e.download_s16(4)
              if not np.array_equal(o1, o2): issues.append("two runs differ")
              land = orc.ebur128(o1.astype(np.float64) / 32768.0, 44100, True, True)
              if abs(land["integrated"] - r1.output_lufs) > 0.011: issues.append("landing")
          except L.JtError as ex:
              if ex.code not in (L.JT_E_SILENT,): issues.append(f"job failed {ex}")
      bad += bool(issues)
      if issues: failures.append((c, what, sr, ch, issues))
      if verbose: print(f"case {c:3d} {what:28s} {sr} Hz x{ch} {secs:5.1f} s cadence {fs}{'v' if var else ''} intervals {niv}: {'ok' if not issues else 'FAILED: ' + '; '.join(issues)}", flush=True)
  return bad, failures


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    bad, failures = run(Engine(0), cases, seed)
    print(f"{cases} cases, {bad} failed", failures)
