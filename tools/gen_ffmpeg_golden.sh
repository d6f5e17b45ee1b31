#!/bin/bash
# Pins the CPU oracle (and, through it, the GPU path) to the real thing: runs every FFmpeg filter the reference instantiates
# (filters.go:607-962, normalise.go:257-264,446-480,1231-1334, analyser_bands.go:33, analyser_output.go:18)
# with the ffmpeg CLI on deterministic fixtures and stores inputs, outputs and a manifest under tests/golden/ffmpeg/.
# tests/test_ffmpeg_golden.py then compares oracle/ (always) and the HIP kernels (-m gpu) against them; without the vectors
# it skips, loudly.  The build container has no ffmpeg: run this once on any host with ffmpeg >= 8.0 (the reference bundles
# FFmpeg 8.1, docs/Spectral-Metrics-Reference.md:5) and commit tests/golden/ffmpeg/.
#
#   tools/gen_ffmpeg_golden.sh [path/to/ffmpeg]
set -euo pipefail
cd "$(dirname "$0")/.."
FFMPEG="${1:-ffmpeg}"
command -v "$FFMPEG" >/dev/null || { echo "ffmpeg not found (pass its path as the first argument)" >&2; exit 2; }
ver="$("$FFMPEG" -hide_banner -version | head -1)"
echo "using: $ver"
case "$ver" in
  *"version 8."*|*"version n8."*|*"version N-"*) ;;
  *) echo "WARNING: the reference bundles FFmpeg 8.1; vectors from another major version may differ (af_afftdn, af_adeclick were reworked in 5.x/6.x)" >&2 ;;
esac
exec python3 tools/gen_ffmpeg_golden.py --ffmpeg "$FFMPEG" --version "$ver" --out tests/golden/ffmpeg
