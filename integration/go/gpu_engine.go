//go:build jtgpu

// gpu_engine.go — the cgo binding a maintainer adds to internal/processor to run the four-pass path on libjtgpu.so.
//
// Two seams, both complete (no method is a placeholder):
//
//   1. The per-file API seam the CLI injects (cmd/jivetalking/pool.go:85-91 workerPoolDeps.processAudio,
//      analysispool.go:28-38): ProcessAudioGPU / AnalyseOnlyDetailedGPU have the signatures of ProcessAudio /
//      AnalyseOnlyDetailed (processor.go:29,78) and return the same result types, filled from jt_process_result.
//   2. The engine seam inside the package (frame_processor.go:64,164; loudnormDeps, normalise.go:172-188): gpuEngine's
//      upload / analyse / bandRMS / filter / regionSamples / measure / normalise / download replace the runFilterGraph sweeps
//      one by one, for a caller that keeps the scalar control logic in Go.  tools/granular_harness.cpp walks exactly this
//      sequence in C++ and is tested equal to the one-call path (tests/test_gpu_round2.py).
//
// This repository's build image has no Go toolchain, so this file is source only; the C side it binds is what the tests
// exercise through ctypes and through the C++ harness.  Place it in internal/processor/, vendor include/ and
// jivetalking_amd/lib/ under third_party/jtgpu/, build with `-tags jtgpu`, and point defaultWorkerPoolDeps at ProcessAudioGPU.
package processor

/*
#cgo CFLAGS:  -I${SRCDIR}/../../third_party/jtgpu/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/jtgpu/lib -ljtgpu -Wl,-rpath,${SRCDIR}/../../third_party/jtgpu/lib
#include "jtgpu.h"
#include "jt_host.h"
#include <stdlib.h>
#include <string.h>

extern void jtProgressTrampoline(void *user, jt_progress_update *u);
static jt_progress_fn jt_trampoline_ptr(void) { return (jt_progress_fn)jtProgressTrampoline; }
*/
import "C"

import (
	"context"
	"fmt"
	"math"
	"os"
	"path/filepath"
	"runtime/cgo"
	"strconv"
	"sync"
	"sync/atomic"
	"time"
	"unsafe"
)

// ROCclr reads GPU_MAX_HW_QUEUES once, when the process first touches HIP.  A handle drives eight streams; with 8 hardware queues each
// has one to itself (DESIGN.md: files in flight).  libjtgpu.so neither reads nor writes the environment (getenv / setenv beside running
// goroutines is a data race), so the host sets it here, in init(), before cgo can have called into HIP; an operator's own value wins.
func init() {
	if _, ok := os.LookupEnv("GPU_MAX_HW_QUEUES"); !ok {
		os.Setenv("GPU_MAX_HW_QUEUES", "8")
	}
}

// gpuDeviceCounter hands worker goroutines their GPU round-robin (files shard one per GPU; no exchange between them).
var (
	gpuDeviceCounter atomic.Int64
	gpuDeviceCount   = 1 // set from the CLI (e.g. rocm-smi count or JIVETALKING_GPUS)
)

// gpuEngine owns one jt_ctx.  Handles are expensive (streams, events, multi-GB device buffers and pinned arenas allocated by the
// first file), so a worker goroutine borrows one from its device's free list and returns it: a handle is created once per
// concurrent worker, not once per file (jt_process_file resets the per-job state itself).
type gpuEngine struct {
	h      *C.jt_ctx
	device int
	mu     sync.Mutex // guards h and gen against a cancel callback that is still running when the job ends
	gen    uint64     // job generation: a callback armed for job g must not cancel job g+1 on the same handle
	closed bool
	lastRC C.int // the code of the last failed engine call (err records it): what finish decides on for the granular seam
}

var (
	gpuPoolMu sync.Mutex
	gpuPool   = map[int][]*gpuEngine{} // device -> idle handles
)

// gpuWorkersPerDevice is how many ProcessAudioGPU calls the CLI keeps in flight per GPU (SetGPUWorkersPerDevice, called where
// runBoundedPool sizes its pool: workers / GPUDeviceCount()).  Three or more handles on one GPU are opened with ONE HIP stream each and
// polling host waits (jt_open_ex): the runtime multiplexes every stream of the process onto a handful of hardware queues, and with
// eight streams per handle one file's event waits stall the other files' streams (ten-minute files, eight in flight: 10 ms per file
// against 15-100), while spinning waits would hold one host core per handle.  One or two handles keep a stream per chain (jt_open).
var gpuWorkersPerDevice atomic.Int32

// SetGPUWorkersPerDevice tells the engine layer how many files the caller keeps in flight per GPU.  Handles already open keep the
// mode they were opened in; call it before the pool starts (or CloseGPUEngines first).
func SetGPUWorkersPerDevice(n int) { gpuWorkersPerDevice.Store(int32(n)) }

func openGPUEngine(device int) (*gpuEngine, error) {
	var h *C.jt_ctx
	streams, flags := C.int(0), C.int(0)
	if gpuWorkersPerDevice.Load() >= 3 {
		streams, flags = 1, C.JT_OPEN_BLOCKING_SYNC
	}
	if rc := C.jt_open_ex(C.int(device), streams, flags, &h); rc != C.JT_OK {
		return nil, fmt.Errorf("jt_open_ex(device %d): code %d", device, int(rc))
	}
	return &gpuEngine{h: h, device: device}, nil
}

// acquireGPUEngine takes an idle handle of the next device (round-robin) or opens one.
func acquireGPUEngine() (*gpuEngine, error) {
	device := int(gpuDeviceCounter.Add(1)-1) % gpuDeviceCount
	gpuPoolMu.Lock()
	if idle := gpuPool[device]; len(idle) > 0 {
		e := idle[len(idle)-1]
		gpuPool[device] = idle[:len(idle)-1]
		gpuPoolMu.Unlock()
		return e, nil
	}
	gpuPoolMu.Unlock()
	return openGPUEngine(device)
}

// finish ends a job: a handle whose job ended with anything but success or a cancellation (JT_E_HIP, an allocation failure, ...) is
// closed, not pooled - the next file then opens a fresh one, as every file did before handles were pooled, and one bad handle cannot
// fail the files after it.
func (e *gpuEngine) finish(rc C.int) {
	if rc == C.JT_OK || rc == C.JT_E_CANCELLED || rc == C.JT_E_INVAL || rc == C.JT_E_UNSUPPORTED || rc == C.JT_E_SILENT {
		e.release() // (bad input / refused request / silent audio: the handle itself is fine)
		return
	}
	e.Close()
}

// release ends the job (late cancel callbacks of this job become no-ops) and returns the handle to its device's free list.
func (e *gpuEngine) release() {
	e.mu.Lock()
	e.gen++
	C.jt_end_job(e.h)
	e.mu.Unlock()
	gpuPoolMu.Lock()
	gpuPool[e.device] = append(gpuPool[e.device], e)
	gpuPoolMu.Unlock()
}

// Close destroys the handle (CloseGPUEngines when the worker pool has drained; finish closes a handle whose job failed).
func (e *gpuEngine) Close() {
	e.mu.Lock()
	if !e.closed {
		e.closed = true
		e.gen++
		C.jt_close(e.h)
	}
	e.mu.Unlock()
}

// CloseGPUEngines closes every idle handle and with them their multi-GB device buffers and pinned arenas.  Wire it where the CLI's
// pools end: `defer processor.CloseGPUEngines()` next to runBoundedPool's wg.Wait() (cmd/jivetalking/pool.go:122-153) and in
// runAnalysisPool - INTEGRATION.md shows both lines.  Nothing else frees pooled handles.
func CloseGPUEngines() {
	gpuPoolMu.Lock()
	defer gpuPoolMu.Unlock()
	for d, idle := range gpuPool {
		for _, e := range idle {
			e.Close()
		}
		delete(gpuPool, d)
	}
}

// err maps a C return code onto the reference's error conventions: ctx cancellation is returned as ctx.Err()
// (frame_processor.go:116-118), everything else as a wrapped message.
func (e *gpuEngine) err(ctx context.Context, rc C.int, what string) error {
	if rc != C.JT_OK {
		e.lastRC = rc
	}
	switch rc {
	case C.JT_OK:
		return nil
	case C.JT_E_CANCELLED:
		if ctx != nil && ctx.Err() != nil {
			return ctx.Err()
		}
		return context.Canceled
	case C.JT_E_SILENT:
		return fmt.Errorf("%s: %s", what, C.GoString(C.jt_last_error(e.h))) // "cannot normalise silent audio" (normalise.go:840-842)
	}
	return fmt.Errorf("%s: %s", what, C.GoString(C.jt_last_error(e.h)))
}

// watch brackets a job and arms jt_cancel for it.  jt_begin_job clears the sticky flag BEFORE the callback is armed and keeps the
// job's own entry points from clearing it again, so a ctx cancelled between here and the first C call is observed by that call.
// context.AfterFunc's stop() does not wait for a callback that has already started; the callback therefore takes e.mu and checks
// the generation, and release / Close bump the generation under the same mutex: a late callback can neither touch a closed
// handle nor cancel the next job on a reused one.
func (e *gpuEngine) watch(ctx context.Context) (stop func() bool) {
	e.mu.Lock()
	gen := e.gen
	C.jt_begin_job(e.h)
	e.mu.Unlock()
	return context.AfterFunc(ctx, func() {
		e.mu.Lock()
		if !e.closed && e.gen == gen {
			C.jt_cancel(e.h)
		}
		e.mu.Unlock()
	})
}

// ------------------------------------------------------------------------------------------------------------------
// Seam 1: ProcessAudio / AnalyseOnlyDetailed
// ------------------------------------------------------------------------------------------------------------------

type progressBridge struct {
	cb  ProgressCallback
	cfg *EffectiveFilterConfig // filled when the Pass-2 start event arrives, so the TUI gets a Go struct
}

//export jtProgressTrampoline
func jtProgressTrampoline(user unsafe.Pointer, u *C.jt_progress_update) {
	b := cgo.Handle(uintptr(user)).Value().(*progressBridge)
	if b.cb == nil {
		return
	}
	up := ProgressUpdate{
		Pass:     PassNumber(u.pass),
		PassName: C.GoString(u.pass_name),
		Progress: float64(u.progress),
		Level:    float64(u.level),
		Duration: float64(u.duration),
	}
	if u.measurements != nil {
		up.Measurements = measurementsFromC(u.measurements, nil)
	}
	if u.config != nil {
		cfg := effectiveConfigFromC(u.config)
		up.Config = cfg
		if u.diag != nil {
			up.Diagnostics = diagnosticsFromC(u.diag, u.measurements)
		}
	}
	if u.has_limiter != 0 {
		up.Limiter = &LimiterProgress{Enabled: u.limiter_enabled != 0, Ceiling: float64(u.limiter_ceiling)}
	}
	b.cb(up)
}

// ProcessAudioGPU is the drop-in for ProcessAudio (processor.go:78): same arguments, same result, same error behaviour
// (wrapped errors, ctx.Err() on cancellation, no temp residue: jt_process_file writes a hidden sibling and renames).
func ProcessAudioGPU(ctx context.Context, inputPath string, config *BaseFilterConfig, progressCallback ProgressCallback) (*ProcessingResult, error) {
	if err := ctx.Err(); err != nil {
		return nil, err
	}
	e, err := acquireGPUEngine()
	if err != nil {
		return nil, err
	}
	var rc C.int = C.JT_OK
	defer func() { e.finish(rc) }()
	stop := e.watch(ctx)
	defer stop()
	if err := validateAdeclickMethod(config.Adeclick.Method); err != nil {
		return nil, err
	}

	base := hostConfigToC(config)
	res := (*C.jt_process_result)(C.calloc(1, C.size_t(unsafe.Sizeof(C.jt_process_result{}))))
	defer C.free(unsafe.Pointer(res))
	bridge := &progressBridge{cb: progressCallback}
	hnd := cgo.NewHandle(bridge)
	defer hnd.Delete()
	cpath := C.CString(inputPath)
	defer C.free(unsafe.Pointer(cpath))
	var outPath [4096]C.char
	var cb C.jt_progress_fn
	if progressCallback != nil {
		cb = C.jt_trampoline_ptr()
	}
	// frame_samples = 0: the file's own decoder-frame cadence (the stream's FLAC block size(s), the WAV demuxer's 4096-byte packets) --
	// what Reader.ReadFrame hands the reference's frame loop, which closes its 250 ms intervals (analyser.go:588-600)
	rc = C.jt_process_file(e.h, cpath, &base, 0, C.JT_FLAC_MD5|C.JT_FILE_PROGRESS_TICKS, cb, unsafe.Pointer(uintptr(hnd)), res, &outPath[0], 4096, nil)
	if rc != C.JT_OK {
		what := "processing failed"
		if res.pass_ms[0] == 0 {
			what = "pass 1 failed" // (the reference wraps per pass: processor.go:89,146,180)
		}
		return nil, e.err(ctx, rc, what)
	}
	return processingResultFromC(e, res, C.GoString(&outPath[0])), nil
}

// AnalyseOnlyDetailedGPU is the drop-in for AnalyseOnlyDetailed (processor.go:29-69): Pass 1 + AdaptConfig.
func AnalyseOnlyDetailedGPU(ctx context.Context, inputPath string, config *BaseFilterConfig, progressCallback ProgressCallback) (*AnalysisResult, error) {
	e, err := acquireGPUEngine()
	if err != nil {
		return nil, err
	}
	var rc C.int = C.JT_OK
	defer func() { e.finish(rc) }()
	stop := e.watch(ctx)
	defer stop()
	if progressCallback != nil {
		progressCallback(ProgressUpdate{Pass: PassAnalysis, PassName: "Analysing"})
	}
	start := time.Now()
	if _, err := e.loadFile(ctx, inputPath); err != nil {
		rc = e.lastRC
		return nil, fmt.Errorf("analysis failed: %w", err)
	}
	base := hostConfigToC(config)
	res := (*C.jt_process_result)(C.calloc(1, C.size_t(unsafe.Sizeof(C.jt_process_result{}))))
	defer C.free(unsafe.Pointer(res))
	if rc = C.jt_analyse_only(e.h, &base, 0, res); rc != C.JT_OK { // (0: the loaded file's own frame cadence)
		return nil, fmt.Errorf("analysis failed: %w", e.err(ctx, rc, "pass 1"))
	}
	m := measurementsFromC(&res.input, e)
	analysis := time.Since(start)
	if progressCallback != nil {
		progressCallback(ProgressUpdate{Pass: PassAnalysis, PassName: "Analysing", Progress: 1.0, Duration: m.Duration, Measurements: m})
	}
	return &AnalysisResult{
		Measurements:       m,
		Config:             effectiveConfigFromC(&res.effective),
		Diagnostics:        diagnosticsFromC(&res.diag, &res.input),
		AnalysisDuration:   analysis - time.Duration(float64(res.stage_ms[3])*1e6),
		AdaptationDuration: time.Duration(float64(res.stage_ms[3]) * 1e6),
	}, nil
}

// ------------------------------------------------------------------------------------------------------------------
// Seam 2: the engine calls, one per runFilterGraph sweep
// ------------------------------------------------------------------------------------------------------------------

// loadFile replaces audio.OpenAudioFile + the ReadFrame loop (reader.go:29-169) for FLAC / WAV; other containers keep the
// libavformat reader and hand their PCM to uploadPCM.
func (e *gpuEngine) loadFile(ctx context.Context, path string) (*C.jt_audio_meta, error) {
	img, err := readWholeFile(path)
	if err != nil {
		return nil, fmt.Errorf("failed to open input file: %w", err)
	}
	var m C.jt_audio_meta
	rc := C.jt_load_audio(e.h, (*C.uint8_t)(unsafe.Pointer(&img[0])), C.int64_t(len(img)), &m)
	return &m, e.err(ctx, rc, "decode")
}

// uploadPCM replaces the ReadFrame -> AVBuffersrcAddFrameFlags feed of Pass 1/2 (frame_processor.go:131-146).  bits / isFloat
// describe the decoder's native sample format (it selects the band graphs' arithmetic: include/jtgpu.h jt_set_source_format);
// channelMask is the decoder context's channel layout (AVChannelLayout.u.mask for native-order layouts; 0 when the decoder reports
// none: libswresample then assumes the default layout of the channel count, and so does the library) -- it selects the row of
// libswresample's default matrix that aformat=channel_layouts=mono applies (filters.go:607-615).
func (e *gpuEngine) uploadPCM(ctx context.Context, pcm []float32, sampleRate, channels, bits int, isFloat bool, channelMask uint64) error {
	frames := len(pcm) / channels
	if rc := C.jt_upload_pcm_layout(e.h, (*C.float)(unsafe.Pointer(&pcm[0])), C.int64_t(frames), C.int(sampleRate), C.int(channels), C.uint64_t(channelMask)); rc != C.JT_OK {
		return e.err(ctx, rc, "upload")
	}
	f := C.int(0)
	if isFloat {
		f = 1
	}
	return e.err(ctx, C.jt_set_source_format(e.h, C.int(bits), f), "source format")
}

// analyse replaces collectAnalysisFrames' sweep (analyser.go:538-650): per-decoder-frame sum(x^2) / peak (OnInputFrame,
// analyser_metrics.go:273-358) and one record per 100 ms ebur128 output frame (OnFrame, analyser.go:621-630), then the interval
// series exactly as intervalAccumulator builds it.
//
// frameSamples is what Reader.ReadFrame delivers per call (it closes the 250 ms intervals, analyser.go:588-600): the caller's own
// decoder's frame length after uploadPCM, or 0 after loadFile = the file's own cadence as jt_load_audio recorded it (the FLAC
// stream's block size -- frame by frame when the frames differ in length -- or the WAV demuxer's 4096-byte packets).
func (e *gpuEngine) analyse(ctx context.Context, frameSamples int, totalFrames int64, sampleRate, channels int) (*C.jt_analysis, []C.jt_interval, error) {
	var lens []C.int32_t
	fs := C.int(frameSamples)
	nFrames := int64(0)
	if frameSamples == 0 {
		var variable C.int
		var nf C.int64_t
		if rc := C.jt_input_frame_layout(e.h, &fs, &variable, &nf, nil, 0); rc != C.JT_OK {
			return nil, nil, e.err(ctx, rc, "frame layout")
		}
		nFrames = int64(nf)
		if variable != 0 {
			lens = make([]C.int32_t, nFrames)
			if rc := C.jt_input_frame_layout(e.h, nil, nil, nil, &lens[0], C.int64_t(nFrames)); rc != C.JT_OK {
				return nil, nil, e.err(ctx, rc, "frame layout")
			}
		}
	} else {
		nFrames = (totalFrames + int64(frameSamples) - 1) / int64(frameSamples)
	}
	nMeta := totalFrames/int64(sampleRate/10) + 2
	a := new(C.jt_analysis)
	ss := make([]C.double, nFrames)
	pk := make([]C.double, nFrames)
	meta := make([]C.jt_frame_meta, nMeta)
	rc := C.jt_pass1(e.h, C.int(frameSamples), a, &ss[0], &pk[0], C.int64_t(nFrames), &meta[0], C.int64_t(nMeta))
	if rc != C.JT_OK {
		return nil, nil, e.err(ctx, rc, "pass 1")
	}
	iv := make([]C.jt_interval, totalFrames/int64(sampleRate/5)+16)
	var lp *C.int32_t
	if len(lens) > 0 {
		lp = &lens[0]
	}
	n := C.jt_host_build_intervals_v(C.int(sampleRate), C.int64_t(totalFrames), fs, lp, C.int(channels), &ss[0], &pk[0], C.int64_t(nFrames),
		&meta[0], a.n_frames_meta, 1, &iv[0], C.int64_t(len(iv)))
	return a, iv[:int(n)], nil
}

// bandRMS replaces measureSpeechBandRMS's region graph (analyser_bands.go:43-95) for all bands of one region.
func (e *gpuEngine) bandRMS(ctx context.Context, start, duration time.Duration, lo, hi []float64) ([]float64, []bool, error) {
	out := make([]float64, len(lo))
	ok := make([]C.int, len(lo))
	rc := C.jt_band_rms(e.h, C.double(fmtSeconds(start)), C.double(fmtSeconds(duration)),
		(*C.double)(unsafe.Pointer(&lo[0])), (*C.double)(unsafe.Pointer(&hi[0])), C.int(len(lo)), (*C.double)(unsafe.Pointer(&out[0])), &ok[0])
	found := make([]bool, len(lo))
	for i := range ok {
		found[i] = ok[i] != 0
		out[i] = parsedFloat("%f", out[i]) // astats prints "%f"
	}
	return out, found, e.err(ctx, rc, "band rms")
}

// filter replaces processWithFilters' sweep (processor.go:255-373).
func (e *gpuEngine) filter(ctx context.Context, cfg *EffectiveFilterConfig) (*OutputMeasurements, error) {
	hc := effectiveConfigToC(cfg)
	var p C.jt_filter_params
	C.jt_host_filter_params(&hc, &p) // the numbers at the precision BuildFilterSpec prints (filters.go:755,811,844,883,906,927)
	var out C.jt_analysis
	if rc := C.jt_pass2(e.h, &p, &out); rc != C.JT_OK {
		return nil, e.err(ctx, rc, "pass 2")
	}
	return outputMeasurementsFromC(&out), nil
}

// regionSamples replaces MeasureOutputRegions (analyser_output.go:276-317): room tone and speech region of one stage output.
func (e *gpuEngine) regionSamples(ctx context.Context, stage int, roomTone *NoiseProfile, speech *SpeechCandidateMetrics) (rt, sp *RegionSample, err error) {
	var st, du [2]C.double
	if roomTone != nil && roomTone.Duration > 0 {
		st[0], du[0] = C.double(fmtSeconds(roomTone.Start)), C.double(fmtSeconds(roomTone.Duration))
	}
	if speech != nil && speech.Region.Duration > 0 {
		st[1], du[1] = C.double(fmtSeconds(speech.Region.Start)), C.double(fmtSeconds(speech.Region.Duration))
	}
	if du[0] <= 0 && du[1] <= 0 {
		return nil, nil, nil
	}
	var out [2]C.jt_region_sample
	if rc := C.jt_region_measure_pair(e.h, C.int(stage), &st[0], &du[0], &out[0]); rc != C.JT_OK {
		return nil, nil, e.err(ctx, rc, "region measure")
	}
	if du[0] > 0 {
		rt = regionSampleFromC(&out[0])
	}
	if du[1] > 0 {
		sp = regionSampleFromC(&out[1])
	}
	return rt, sp, nil
}

func limiterPlanToC(lim limiterPlan) C.jt_limiter_plan {
	var p C.jt_limiter_plan
	if lim.needed {
		p.needed = 1
		p.pre_gain_db = C.double(parsedFloat("%.1f", math.Max(lim.preGainDB, 0))) // volume=%.1fdB
		p.limit = C.double(parsedFloat("%.6f", Decibels(lim.ceilingDB).LinearAmplitude().Float64()))
	}
	return p
}

// measure replaces measureWithLoudnorm (normalise.go:226-346): the values arrive as doubles and are rounded to the "%.2f"
// strings loudnorm's JSON carries before anything consumes them.
func (e *gpuEngine) measure(ctx context.Context, lim limiterPlan, ln LoudnormConfig) (*LoudnormMeasurement, error) {
	p := limiterPlanToC(lim)
	var s C.jt_loudnorm_stats
	if rc := C.jt_pass3(e.h, &p, C.double(ln.TargetI), C.double(ln.TargetTP), C.double(ln.TargetLRA), &s); rc != C.JT_OK {
		return nil, e.err(ctx, rc, "pass 3")
	}
	m := &LoudnormMeasurement{
		InputI:      parsedFloat("%.2f", float64(s.input_i)),
		InputTP:     parsedFloat("%.2f", float64(s.input_tp)),
		InputLRA:    parsedFloat("%.2f", float64(s.input_lra)),
		InputThresh: parsedFloat("%.2f", float64(s.input_thresh)),
	}
	m.TargetOffset = ln.TargetI - m.InputI // (not consumed: normalise.go:861-873 derives its own offset)
	return m, nil
}

// normalise replaces applyLoudnormAndMeasure (normalise.go:924-1190): volume / alimiter prefix, loudnorm (linear), adeclick,
// brickwall alimiter, analysis, s16.  sourceRate is the Pass-2 output rate (44100).
func (e *gpuEngine) normalise(ctx context.Context, lim limiterPlan, m *LoudnormMeasurement, offset float64, cfg *EffectiveFilterConfig, sourceRate int) (*OutputMeasurements, *LoudnormStats, error) {
	hc := effectiveConfigToC(cfg)
	var ms C.jt_loudnorm_stats
	ms.input_i, ms.input_tp, ms.input_lra, ms.input_thresh = C.double(m.InputI), C.double(m.InputTP), C.double(m.InputLRA), C.double(m.InputThresh)
	var dec C.jt_limiter_decision
	dec.pre_gain_db, dec.ceiling_db = C.double(lim.preGainDB), C.double(lim.ceilingDB)
	if lim.needed {
		dec.needed = 1
	}
	var ap C.jt_loudnorm_apply
	var spec [2048]C.char
	C.jt_host_pass4_spec(&hc, &ms, C.double(offset), &dec, C.int(sourceRate), nil, &spec[0], 2048, &ap) // numeric content of buildLoudnormFilterSpec
	p := limiterPlanToC(lim)
	var out C.jt_analysis
	var st C.jt_loudnorm_stats
	if rc := C.jt_pass4(e.h, &p, &ap, &out, &st); rc != C.JT_OK {
		return nil, nil, e.err(ctx, rc, "pass 4")
	}
	return outputMeasurementsFromC(&out), loudnormStatsFromC(&st), nil
}

// download replaces the frames handed to Encoder.WriteFrame (encoder.go:145): s16 mono at the output rate.
func (e *gpuEngine) download(ctx context.Context, stage int) ([]int16, error) {
	var n C.int64_t
	C.jt_output_len(e.h, C.int(stage), &n)
	if n <= 0 {
		return nil, fmt.Errorf("no stage %d output on the device", stage)
	}
	pcm := make([]int16, int(n))
	rc := C.jt_download_s16(e.h, C.int(stage), (*C.int16_t)(unsafe.Pointer(&pcm[0])), n, &n)
	return pcm, e.err(ctx, rc, "download")
}

// writeFLAC replaces createOutputEncoder + Encoder.WriteFrame/Flush/Close (encoder.go:54-215) for a stage output.
func (e *gpuEngine) writeFLAC(ctx context.Context, stage int, path string) error {
	var data *C.uint8_t
	var n C.int64_t
	var info C.jt_flac_info
	if rc := C.jt_flac_encode(e.h, C.int(stage), C.JT_FLAC_MD5, &data, &n, &info); rc != C.JT_OK {
		return e.err(ctx, rc, "flac encode")
	}
	tmp, err := createSiblingTempPath(path, "processing")
	if err != nil {
		return err
	}
	if err := writeWholeFile(tmp, unsafe.Slice((*byte)(unsafe.Pointer(data)), int(n))); err != nil {
		removeQuietly(tmp)
		return err
	}
	if err := publishOutput(tmp, path); err != nil {
		removeQuietly(tmp)
		return err
	}
	return nil
}

// ------------------------------------------------------------------------------------------------------------------
// Struct conversions (field for field; the C structs mirror the Go ones: include/jt_host.h)
// ------------------------------------------------------------------------------------------------------------------

func b2i(b bool) C.int {
	if b {
		return 1
	}
	return 0
}
func fmtSeconds(d time.Duration) float64 { return parsedFloat("%f", d.Seconds()) } // regions travel through "%f"-formatted options
func parsedFloat(verb string, v float64) float64 {
	f, err := strconv.ParseFloat(fmt.Sprintf(verb, v), 64)
	if err != nil {
		return v
	}
	return f
}
func linToDB(v float64) float64 {
	if v <= 0 {
		return math.Inf(-1)
	}
	return 20 * math.Log10(v)
}

func biquadToC(b BiquadFilterConfig) C.jt_biquad_cfg {
	return C.jt_biquad_cfg{enabled: b2i(b.Enabled), frequency: C.double(b.Frequency), poles: C.int(b.Poles), width: C.double(b.Width),
		mix: C.double(b.Mix), transform_tdii: b2i(b.Transform == "tdii")}
}
func biquadFromC(b *C.jt_biquad_cfg) BiquadFilterConfig {
	t := ""
	if b.transform_tdii != 0 {
		t = "tdii"
	}
	return BiquadFilterConfig{Enabled: b.enabled != 0, Frequency: float64(b.frequency), Poles: int(b.poles), Width: float64(b.width), Mix: float64(b.mix), Transform: t}
}

func defaultsToC(d *filterConfigDefaults) C.jt_host_config {
	var c C.jt_host_config
	c.downmix_enabled, c.analysis_enabled = b2i(d.Downmix.Enabled), b2i(d.Analysis.Enabled)
	c.resample_enabled, c.resample_rate, c.resample_frame = b2i(d.Resample.Enabled), C.int(d.Resample.SampleRate), C.int(d.Resample.FrameSize)
	c.rumble_hp, c.bandlimit_lp = biquadToC(d.RumbleHighPass), biquadToC(d.BandlimitLowPass)
	n := d.NoiseReduction
	c.nr_enabled, c.nr_strength, c.nr_patch_s, c.nr_research_s, c.nr_smooth = b2i(n.Enabled), C.double(n.Strength), C.double(n.PatchSec), C.double(n.ResearchSec), C.double(n.Smooth)
	c.afftdn_enabled, c.afftdn_nr, c.afftdn_custom, c.afftdn_track_noise = b2i(n.AfftdnEnabled), C.double(n.AfftdnNoiseReduction), b2i(n.AfftdnNoiseType == "custom"), b2i(n.AfftdnTrackNoise)
	c.afftdn_noise_floor = C.double(n.AfftdnNoiseFloor)
	bn := []byte(n.AfftdnBandNoise)
	for i := 0; i < len(bn) && i < 255; i++ {
		c.afftdn_band_noise[i] = C.char(bn[i])
	}
	g := d.SpeechGate
	c.gate_enabled, c.gate_threshold, c.gate_ratio, c.gate_attack, c.gate_release = b2i(g.Enabled), C.double(g.Threshold), C.double(g.Ratio), C.double(g.Attack), C.double(g.Release)
	c.gate_range, c.gate_knee, c.gate_makeup, c.gate_detection_set = C.double(g.Range), C.double(g.Knee), C.double(g.Makeup), b2i(g.Detection != "")
	k := d.LevellingCompressor
	c.comp_enabled, c.comp_threshold_db, c.comp_ratio, c.comp_attack, c.comp_release = b2i(k.Enabled), C.double(k.Threshold), C.double(k.Ratio), C.double(k.Attack), C.double(k.Release)
	c.comp_makeup_db, c.comp_knee, c.comp_mix = C.double(k.Makeup), C.double(k.Knee), C.double(k.Mix)
	s := d.Deesser
	c.deess_enabled, c.deess_intensity, c.deess_amount, c.deess_frequency = b2i(s.Enabled), C.double(s.Intensity), C.double(s.Amount), C.double(s.Frequency)
	a := d.Adeclick
	c.adeclick_enabled, c.adeclick_threshold, c.adeclick_window, c.adeclick_overlap = b2i(a.Enabled), C.double(a.Threshold), C.double(a.Window), C.double(a.Overlap)
	c.adeclick_method_s = C.int(adeclickMethodCode(a.Method))
	l := d.Loudnorm
	c.loudnorm_enabled, c.target_i, c.target_tp, c.target_lra, c.dual_mono, c.linear = b2i(l.Enabled), C.double(l.TargetI), C.double(l.TargetTP), C.double(l.TargetLRA), b2i(l.DualMono), b2i(l.Linear)
	return c
}
// adeclickMethodNames: AdeclickConfig.Method goes into the filter spec verbatim (filters.go:958-960), and af_adeclick's option table
// names each method twice ("a" / "add", "s" / "save").  The code keeps the spelling so that the spec text and the effective config
// round-trip; "" leaves the option out (af_adeclick.c's default, overlap-add).  Anything else would make FFmpeg fail the graph:
// validateAdeclickMethod reports it before a job starts instead of running a different method silently.
var adeclickMethodNames = map[int]string{0: "", 1: "s", 2: "a", 3: "save", 4: "add"}

func adeclickMethodCode(method string) int {
	for code, name := range adeclickMethodNames {
		if name == method {
			return code
		}
	}
	return -1
}

func validateAdeclickMethod(method string) error {
	if adeclickMethodCode(method) < 0 {
		return fmt.Errorf("adeclick: unknown method %q (af_adeclick accepts a, add, s, save)", method)
	}
	return nil
}

func hostConfigToC(cfg *BaseFilterConfig) C.jt_host_config { return defaultsToC(&cfg.filterConfigDefaults) }
func effectiveConfigToC(cfg *EffectiveFilterConfig) C.jt_host_config {
	d := filterConfigDefaults(*cfg)
	return defaultsToC(&d)
}

func effectiveConfigFromC(c *C.jt_host_config) *EffectiveFilterConfig {
	nt := "w"
	if c.afftdn_custom != 0 {
		nt = "custom"
	}
	det := ""
	if c.gate_detection_set != 0 {
		det = "rms"
	}
	method := adeclickMethodNames[int(c.adeclick_method_s)]
	d := filterConfigDefaults{
		Downmix:          DownmixConfig{Enabled: c.downmix_enabled != 0},
		Analysis:         AnalysisConfig{Enabled: c.analysis_enabled != 0},
		Resample:         ResampleConfig{Enabled: c.resample_enabled != 0, SampleRate: int(c.resample_rate), Format: "s16", FrameSize: int(c.resample_frame)},
		RumbleHighPass:   biquadFromC(&c.rumble_hp),
		BandlimitLowPass: biquadFromC(&c.bandlimit_lp),
		NoiseReduction: NoiseReductionConfig{Enabled: c.nr_enabled != 0, Strength: float64(c.nr_strength), PatchSec: float64(c.nr_patch_s),
			ResearchSec: float64(c.nr_research_s), Smooth: float64(c.nr_smooth), AfftdnEnabled: c.afftdn_enabled != 0,
			AfftdnNoiseReduction: float64(c.afftdn_nr), AfftdnNoiseType: nt, AfftdnTrackNoise: c.afftdn_track_noise != 0,
			AfftdnNoiseFloor: float64(c.afftdn_noise_floor), AfftdnBandNoise: C.GoString(&c.afftdn_band_noise[0])},
		SpeechGate: SpeechGateConfig{Enabled: c.gate_enabled != 0, Threshold: float64(c.gate_threshold), Ratio: float64(c.gate_ratio),
			Attack: float64(c.gate_attack), Release: float64(c.gate_release), Range: float64(c.gate_range), Knee: float64(c.gate_knee),
			Makeup: float64(c.gate_makeup), Detection: det},
		LevellingCompressor: LevellingCompressorConfig{Enabled: c.comp_enabled != 0, Threshold: float64(c.comp_threshold_db), Ratio: float64(c.comp_ratio),
			Attack: float64(c.comp_attack), Release: float64(c.comp_release), Makeup: float64(c.comp_makeup_db), Knee: float64(c.comp_knee), Mix: float64(c.comp_mix)},
		Deesser:  DeesserConfig{Enabled: c.deess_enabled != 0, Intensity: float64(c.deess_intensity), Amount: float64(c.deess_amount), Frequency: float64(c.deess_frequency)},
		Adeclick: AdeclickConfig{Enabled: c.adeclick_enabled != 0, Threshold: float64(c.adeclick_threshold), Window: float64(c.adeclick_window), Overlap: float64(c.adeclick_overlap), Method: method},
		Loudnorm: LoudnormConfig{Enabled: c.loudnorm_enabled != 0, TargetI: float64(c.target_i), TargetTP: float64(c.target_tp), TargetLRA: float64(c.target_lra),
			DualMono: c.dual_mono != 0, Linear: c.linear != 0},
		FilterOrder: append([]FilterID(nil), Pass2FilterOrder...),
	}
	e := EffectiveFilterConfig(d)
	return &e
}

func diagnosticsFromC(d *C.jt_adaptive_diag, m *C.jt_measurements) *AdaptiveDiagnostics {
	out := &AdaptiveDiagnostics{
		BandlimitLPReason:             "20.5 kHz band-limit (always on)",
		SpeechGateQuietSpeechEstimate: float64(d.gate_quiet_speech_estimate),
		SpeechGateSpeechSeparation:    float64(d.gate_separation),
		SpeechGateSpeechHeadroom:      float64(d.gate_speech_headroom),
		SpeechGateThresholdUnclamped:  float64(d.gate_threshold_unclamped),
		SpeechGateDepthDB:             float64(d.gate_depth_db),
		SpeechGateNarrowGap:           d.gate_narrow_gap != 0,
		AfftdnEnabled:                 d.afftdn_enabled != 0,
		AfftdnNoiseFloorDB:            float64(d.afftdn_noise_floor_db),
	}
	if m != nil && m.has_speech_profile != 0 {
		out.SpeechGateClampReason = "none"
		if d.gate_narrow_gap != 0 {
			out.SpeechGateClampReason = "narrow_gap"
		}
	}
	switch {
	case d.afftdn_disabled_voice_activated != 0:
		out.AfftdnDisableReason = "voice_activated"
	case d.afftdn_enabled != 0:
		out.AfftdnNoiseType = "w"
		if d.afftdn_custom != 0 {
			out.AfftdnNoiseType = "custom"
		}
	}
	return out
}

func spectralFromC(s *C.jt_spectral) SpectralMetrics {
	return SpectralMetrics{Mean: float64(s.mean), Variance: float64(s.variance), Centroid: float64(s.centroid), Spread: float64(s.spread),
		Skewness: float64(s.skewness), Kurtosis: float64(s.kurtosis), Entropy: float64(s.entropy), Flatness: float64(s.flatness), Crest: float64(s.crest),
		Flux: float64(s.flux), Slope: float64(s.slope), Decrease: float64(s.decrease), Rolloff: float64(s.rolloff), Found: true}
}

func dynamicsFromC(a *C.jt_astats) DynamicsMetrics {
	return DynamicsMetrics{DynamicRange: float64(a.dynamic_range), RMSLevel: float64(a.rms_level), PeakLevel: float64(a.peak_level), RMSTrough: float64(a.rms_trough),
		RMSPeak: float64(a.rms_peak), DCOffset: float64(a.dc_offset), FlatFactor: float64(a.flat_factor), CrestFactor: float64(a.crest_factor),
		ZeroCrossingsRate: float64(a.zero_crossings_rate), ZeroCrossings: float64(a.zero_crossings), MaxDifference: float64(a.max_difference),
		MinDifference: float64(a.min_difference), MeanDifference: float64(a.mean_difference), RMSDifference: float64(a.rms_difference), Entropy: float64(a.entropy),
		MinLevel: float64(a.min_level), MaxLevel: float64(a.max_level), NoiseFloorCount: float64(a.noise_floor_count), BitDepth: float64(a.bit_depth),
		NumberOfSamples: float64(a.number_of_samples)}
}

func regionMetricsSample(r *C.jt_region_metrics) RegionSample {
	return RegionSample{RMSLevel: float64(r.rms_level), PeakLevel: float64(r.peak_level), CrestFactor: float64(r.crest_factor), Spectral: spectralFromC(&r.spectral),
		MomentaryLUFS: float64(r.momentary_lufs), ShortTermLUFS: float64(r.shortterm_lufs), TruePeak: float64(r.true_peak), SamplePeak: float64(r.sample_peak)}
}

// regionSampleFromC applies measureOutputRegionFromReader's conversions (analyser_output.go:95-227): crest = peak - rms in dB,
// peaks 20 log10 of the last linear value.
func regionSampleFromC(r *C.jt_region_sample) *RegionSample {
	return &RegionSample{RMSLevel: float64(r.rms_level), PeakLevel: float64(r.peak_level), CrestFactor: float64(r.peak_level - r.rms_level),
		Spectral: spectralFromC(&r.spectral), MomentaryLUFS: float64(r.momentary), ShortTermLUFS: float64(r.shortterm),
		TruePeak: linToDB(float64(r.true_peak)), SamplePeak: linToDB(float64(r.sample_peak))}
}

func candidateFromC(c *C.jt_speech_candidate) SpeechCandidateMetrics {
	return SpeechCandidateMetrics{
		Region:       SpeechRegion{Start: time.Duration(c.region.start_ns), End: time.Duration(c.region.end_ns), Duration: time.Duration(c.region.duration_ns)},
		RegionSample: regionMetricsSample(&c.sample), VoicingDensity: float64(c.voicing_density),
		BodyBandRMS: float64(c.body_band_rms), SibBandRMS: float64(c.sib_band_rms), BandsMeasured: c.bands_measured != 0, Score: float64(c.score),
		OriginalStart: time.Duration(c.original_start_ns), OriginalDuration: time.Duration(c.original_duration_ns), WasRefined: c.was_refined != 0}
}

var floorSourceNames = [...]string{"astats", "rms_estimate", "ebur128_estimate", "vad_percentile"}

// measurementsFromC rebuilds AudioMeasurements (analyser.go:232-249).  With an engine, the interval series of the handle's last
// analysis is attached (Regions.IntervalSamples feeds the report's distribution and the .intervals.jsonl sidecar).
func measurementsFromC(m *C.jt_measurements, e *gpuEngine) *AudioMeasurements {
	out := &AudioMeasurements{Spectral: spectralFromC(&m.spectral), Duration: float64(m.duration_s)}
	out.Loudness = InputLoudnessMetrics{
		LoudnessMetrics: LoudnessMetrics{MomentaryLoudness: float64(m.momentary), ShortTermLoudness: float64(m.shortterm), SamplePeak: float64(m.sample_peak)},
		InputI:          float64(m.input_i), InputTP: float64(m.input_tp), InputLRA: float64(m.input_lra), InputThresh: float64(m.input_thresh), TargetOffset: float64(m.target_offset)}
	out.Dynamics = dynamicsFromC(&m.dynamics)
	src := ""
	if int(m.floor_source) >= 0 && int(m.floor_source) < len(floorSourceNames) {
		src = floorSourceNames[m.floor_source]
	}
	out.Noise = NoiseMetrics{Floor: float64(m.floor), FloorSource: src, FloorPrescan: float64(m.floor_prescan), FloorAstats: float64(m.floor_astats),
		RoomToneDetectLevel: float64(m.room_tone_detect_level), VoiceActivated: m.voice_activated != 0, FlooredFraction: float64(m.floored_fraction),
		ReductionHeadroom: float64(m.reduction_headroom)}
	r := &out.Regions
	for i := 0; i < int(m.n_speech_regions); i++ {
		s := &m.speech_regions[i]
		r.SpeechRegions = append(r.SpeechRegions, SpeechRegion{Start: time.Duration(s.start_ns), End: time.Duration(s.end_ns), Duration: time.Duration(s.duration_ns)})
	}
	for i := 0; i < int(m.n_candidates); i++ {
		r.SpeechCandidates = append(r.SpeechCandidates, candidateFromC(&m.candidates[i]))
	}
	if m.has_speech_profile != 0 {
		sp := candidateFromC(&m.speech_profile)
		r.SpeechProfile = &sp
		for i := range r.SpeechCandidates { // "pointer into SpeechCandidates" (analyser.go:209)
			if r.SpeechCandidates[i].Region.Start == sp.Region.Start && r.SpeechCandidates[i].Region.Duration == sp.Region.Duration {
				r.SpeechCandidates[i] = sp
				r.SpeechProfile = &r.SpeechCandidates[i]
				break
			}
		}
	}
	if m.has_noise_profile != 0 {
		p := &m.noise_profile
		np := &NoiseProfile{Start: time.Duration(p.start_ns), Duration: time.Duration(p.duration_ns), MeasuredNoiseFloor: float64(p.measured_noise_floor),
			PeakLevel: float64(p.peak_level), CrestFactor: float64(p.crest_factor), Entropy: float64(p.entropy), Spectral: spectralFromC(&p.spectral),
			BandsMeasured: p.bands_measured != 0}
		for i := 0; i < int(p.band_noise_n); i++ {
			np.BandNoise = append(np.BandNoise, float64(p.band_noise[i]))
		}
		switch p.warning {
		case 1:
			np.ExtractionWarning = fmt.Sprintf("using short room tone region (%.1fs) - ideally need >=%ds", np.Duration.Seconds(), 8)
		case 2:
			np.ExtractionWarning = fmt.Sprintf("using long room tone region (%.1fs) - ideally <=%ds", np.Duration.Seconds(), 18)
		}
		r.NoiseProfile = np
	}
	if m.has_room_tone_sample != 0 {
		s := regionMetricsSample(&m.room_tone_sample)
		r.ElectedRoomToneSample = &s
	}
	r.VoicedLowPercentile, r.NoiseHighPercentile, r.GateSeparationDB = float64(m.voiced_low_percentile), float64(m.noise_high_percentile), float64(m.gate_separation_db)
	if e != nil {
		r.IntervalSamples = e.intervals()
	}
	return out
}

// intervals copies the handle's interval series (IntervalSample, analyser_metrics.go:17-32).
func (e *gpuEngine) intervals() []IntervalSample {
	n := int64(C.jt_host_last_intervals(e.h, nil, 0))
	if n <= 0 {
		return nil
	}
	raw := make([]C.jt_interval, n)
	C.jt_host_last_intervals(e.h, &raw[0], C.int64_t(n))
	out := make([]IntervalSample, n)
	for i := range raw {
		r := &raw[i]
		out[i] = IntervalSample{Timestamp: time.Duration(r.timestamp_ns), RMSLevel: float64(r.rms_level), PeakLevel: float64(r.peak_level),
			Spectral: spectralFromC(&r.spectral), MomentaryLUFS: float64(r.momentary_lufs), ShortTermLUFS: float64(r.shortterm_lufs),
			TruePeak: float64(r.true_peak), SamplePeak: float64(r.sample_peak)}
		out[i].Spectral.Found = r.spectral_found != 0
	}
	return out
}

// small file helpers (os.ReadFile / os.WriteFile with the reference's 0644 output mode, file_write.go:47-53)
func readWholeFile(path string) ([]byte, error) { return os.ReadFile(path) }
func writeWholeFile(path string, b []byte) error { return os.WriteFile(path, b, 0o644) }
func removeQuietly(path string)                  { _ = os.Remove(path) }

// outputMeasurementsFromC applies the Go-side metadata conversions of extractOutputFrameMetadata / finalizeOutputMeasurements
// (analyser_metrics.go:947-1040): ebur128 values as printed ("%.3f"), peaks 20 log10, astats crest linear -> dB, min / max level ->
// dBFS, thresh = I - 10 (f_ebur128.c exports no target_threshold key).
func outputMeasurementsFromC(a *C.jt_analysis) *OutputMeasurements {
	q3 := func(v C.double) float64 { return parsedFloat("%.3f", float64(v)) }
	qf := func(v C.double) float64 { return parsedFloat("%f", float64(v)) }
	om := &OutputMeasurements{Spectral: spectralFromC(&a.spectral_mean)}
	om.Loudness = OutputLoudnessMetrics{
		LoudnessMetrics: LoudnessMetrics{MomentaryLoudness: q3(a.r128.momentary), ShortTermLoudness: q3(a.r128.shortterm), SamplePeak: linToDB(q3(a.r128.sample_peak))},
		OutputI:         q3(a.r128.integrated), OutputTP: linToDB(q3(a.r128.true_peak)), OutputLRA: q3(a.r128.lra)}
	if om.Loudness.OutputI != 0 {
		om.Loudness.OutputThresh = om.Loudness.OutputI - 10.0
	}
	d := dynamicsFromC(&a.astats)
	d.DynamicRange, d.RMSLevel, d.PeakLevel, d.RMSTrough, d.RMSPeak = qf(a.astats.dynamic_range), qf(a.astats.rms_level), qf(a.astats.peak_level), qf(a.astats.rms_trough), qf(a.astats.rms_peak)
	d.CrestFactor = linearRatioToDB(qf(a.astats.crest_factor))
	d.MinLevel, d.MaxLevel = linearSampleToDBFS(qf(a.astats.min_level)), linearSampleToDBFS(qf(a.astats.max_level))
	om.Dynamics = d
	return om
}

func loudnormStatsFromC(s *C.jt_loudnorm_stats) *LoudnormStats {
	f := func(v C.double) string { return fmt.Sprintf("%.2f", float64(v)) }
	t := "linear"
	if s.normalization_type_dynamic != 0 {
		t = "dynamic"
	}
	return &LoudnormStats{InputI: f(s.input_i), InputTP: f(s.input_tp), InputLRA: f(s.input_lra), InputThresh: f(s.input_thresh),
		OutputI: f(s.output_i), OutputTP: f(s.output_tp), OutputLRA: f(s.output_lra), OutputThresh: f(s.output_thresh),
		NormalizationType: t, TargetOffset: f(s.target_offset)}
}

func processingResultFromC(e *gpuEngine, res *C.jt_process_result, outputPath string) *ProcessingResult {
	m := measurementsFromC(&res.input, e)
	out := &ProcessingResult{
		OutputPath:   outputPath,
		InputLUFS:    float64(res.input_lufs),
		OutputLUFS:   float64(res.output_lufs),
		Measurements: m,
		Config:       effectiveConfigFromC(&res.effective),
		Diagnostics:  diagnosticsFromC(&res.diag, &res.input),
		RegionTimings: RegionMeasurementTimings{FilteredOutput: time.Duration(float64(res.stage_ms[5]) * 1e6),
			FinalOutput: time.Duration(float64(res.stage_ms[9]) * 1e6)},
	}
	out.InputMetadata = InputMetadata{DurationSecs: m.Duration} // (SampleRate / Channels: from jt_audio_meta when the caller loaded the file)
	fm := outputMeasurementsFromC(&res.filtered)
	if res.has_region_samples != 0 {
		if res.filtered_room_tone.frames > 0 {
			fm.RoomToneSample = regionSampleFromC(&res.filtered_room_tone)
		}
		if res.filtered_speech.frames > 0 {
			fm.SpeechSample = regionSampleFromC(&res.filtered_speech)
		}
	}
	out.FilteredMeasurements = fm
	if res.effective.loudnorm_enabled != 0 && res.final_.n_frames_meta > 0 {
		final := outputMeasurementsFromC(&res.final_)
		if res.final_room_tone.frames > 0 {
			final.RoomToneSample = regionSampleFromC(&res.final_room_tone)
		}
		if res.final_speech.frames > 0 {
			final.SpeechSample = regionSampleFromC(&res.final_speech)
		}
		stats := loudnormStatsFromC(&res.loudnorm)
		nr := &NormalisationResult{
			InputLUFS: float64(res.measure.input_i), InputTP: float64(res.measure.input_tp),
			OutputLUFS: float64(res.output_lufs), OutputTP: float64(res.output_tp_db),
			GainApplied: float64(res.offset), WithinTarget: res.within_target != 0,
			LoudnormStats: stats, LoudnormParsed: parseLoudnormMeasured(stats, float64(res.effective_target_i)),
			RequestedTargetI: float64(res.effective.target_i), EffectiveTargetI: float64(res.effective_target_i),
			LinearModeForced: res.linear_possible == 0, ActualNormDynamic: res.loudnorm.normalization_type_dynamic != 0,
			LimiterDiagnostics: LimiterDiagnostics{LimiterEnabled: res.limiter.needed != 0, LimiterCeiling: float64(res.limiter.ceiling_db),
				LimiterGain: float64(res.limiter.gain_db), LimiterFilteredTP: float64(res.limiter.filtered_tp), PreGainDB: float64(res.limiter.pre_gain_db),
				LimiterClamped: res.limiter.clamped != 0},
			Pass3FilterPrefix:     C.GoString(&res.limiter.pass3_prefix[0]),
			RegionMeasurementTime: time.Duration(float64(res.stage_ms[9]) * 1e6),
			FinalMeasurements:     final,
		}
		out.NormResult = nr
	}
	_ = filepath.Base // (run record assembly uses filepath.Base(result.OutputPath), runrecord.go:291)
	return out
}
