// fpv_kernels.h -- launch interface between the C ABI (fpv_capi.hip) and the
// gfx950 kernels (fpv_kernels.hip).  Internal; the public boundary is
// include/fpv_hip.h.
#ifndef FPV_KERNELS_H_
#define FPV_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpvhip {

// Number of 256-bin histograms taken per frame by the exact (redo) decision stage.
//   0: raw high bytes at stride 15                     (delta decision, cc:526-531)
//   1/2: a / a - CG at stride 31, delta NOT applied    (CG decision,    cc:554-562)
//   3/4: the same with the delta planes subtracted
constexpr int kNumHists = 5;

struct Geom {
  uint32_t xs, ys, npix;        // frame
  uint32_t pw, ph, npv;         // preview = (xs/4) x (ys/4)
  uint32_t strip_rows, strips;  // streaming kernel: rows per wave strip, strips per frame
  uint32_t redo_strip_rows, redo_strips;  // the same for the redo pass of mis-speculated frames (shorter strips)
  uint32_t redo_rows[4];                  // strip heights the redo pass chooses from by list length (ascending, 0 = none)
  uint32_t ring_fwd, ring_base; // streaming kernel: LDS ring size (pow2), offset of strip pixel 0
  uint32_t nacc;                // streaming kernel: preview accumulator rows
  uint32_t pwr, pvb_stride;     // pw rounded up to 16; bytes of one (frame, strip) boundary record
  uint32_t wave_lds;            // streaming kernel: LDS bytes per wave
  uint32_t stream_waves_per_cu; // streaming kernel: persistent waves per CU
  uint32_t wg_waves;            // streaming kernel: waves (= consecutive frames) per workgroup
  uint32_t fast;                // streaming kernel: the FAST variant applies (xs % 64 == 0, full steps)
  // tile pass (fpv_tile.inc): 0 = the geometry takes the streaming kernels instead
  uint32_t tile_ok, tile_groups, tile_threads, tile_tiles, tile_lds;
  // round 4's tile pass (fpv_tile8.inc: 8 pixels per lane, R rows per thread); preferred when usable
  uint32_t tile8_ok, tile8_rows, tile8_groups, tile8_threads, tile8_tiles, tile8_lds, tile8_frame_big;
  uint32_t tile_lfgroup;     // log2 of the frames per group the tile pass walks tile by tile (large frames; 0: frame by frame)
  uint32_t tile_lfgroup_fixed;  // FPV_TILE_FGROUP given: not shortened to the batch
  uint32_t tile_decide_big;  // FPV_DECIDE_BIG (experiments / tests): k_tile_decide with 1024 threads whatever the size
  uint32_t nd, ncg;             // sample counts of the two heuristics
  uint32_t ring;                // inverse kernel: LDS ring size (power of two)
  uint32_t inv_ring;            // fast inverse kernel: ring size in pixels (power of two)
  uint32_t spec_k4;             // speculative inverse kernel: dwords (4 pixels) per lane and superstep, 0 = unusable
  uint32_t spec_ring;           // speculative inverse kernel: ring size in bytes (power of two)
  int32_t shift, swap;          // Frame ctor arguments (swap = big_endian on LE host)
  int32_t fwd_ok;               // forward kernels usable (dims multiples of 4, LDS fits)
};

// Fills every derived field from xs, ys, shift, big_endian.  Returns false (and
// a message) when the geometry is not supported by the kernels.
bool make_geom(uint32_t xs, uint32_t ys, int shift, int big_endian, Geom* g,
               const char** why);
// Inverse-only geometry: any non-zero dimensions.
bool make_geom_decoder(uint32_t xs, uint32_t ys, Geom* g, const char** why);

struct FwdWorkspace {
  uint32_t* ctr;         // two counter blocks of 64 words (alternating batches): list sizes, task counters, votes
  uint32_t* anylow;      // [nframes]  OR of raw low bytes
  uint32_t* frame_done;  // [nframes]  strips of the frame that have finished
  uint32_t* bdone;       // [nframes][redo_strips] arrivals at the boundary above each strip (parity counts)
  uint8_t* decision;     // [nframes]  bit0 delta, bit1 CG
  uint32_t* list_a;      // frames whose delta decision was mis-speculated
  uint32_t* list_b;      // frames to redo under their true flags
  uint8_t* pvb;          // [nframes][strips] raw first / last preview row of every strip
  uint16_t* partials;    // [nframes][strips][3][256] per-strip histograms of the fused pass
  uint8_t* spec;         // speculated flags for the next batch (majority of the last one)
  uint8_t* trec;         // tile pass: [nframes][tiles] per-tile records (partial heuristic histograms, low-byte word)
  uint8_t* pvraw;        // tile pass: [nframes][npv] raw preview
  uint32_t* spec_pos;    // tile pass: [nframes] per-position prediction state
  uint32_t* dpart;       // tile pass: [kDecideSlots][kDecidePartWords] k_tile_decide's partial sums when several workgroups share a frame
  uint32_t nframes;      // capacity
  size_t zero_bytes;     // leading bytes cleared once at creation (counters, anylow, frame_done, bdone)
  uint32_t seq;          // batches launched so far (selects the counter block); host side
};
size_t fwd_workspace_bytes(const Geom& g, uint32_t nframes);
void carve_fwd_workspace(const Geom& g, uint32_t nframes, void* base,
                         FwdWorkspace* ws);
hipError_t init_fwd_workspace(const FwdWorkspace& ws, hipStream_t s);

// Per-device one-time set-up of the kernels (function attributes); idempotent, thread-safe.
void prepare_kernels();

// Plain device copy (bench: measured HBM ceiling).  shape 1: one 16-byte element per thread,
// dispatch-ordered (the fastest shape found); 0: round 2's chunked persistent copy.
hipError_t launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s, int shape = 1);

// Raw split of one frame into byte planes (delta frame set-up, cc:370-451).
hipError_t launch_split(const Geom& g, const uint16_t* d_img, uint8_t* d_high,
                        uint8_t* d_low, uint32_t* d_anylow, hipStream_t s);
// high << 8 | low for npix_total pixels (multiple of 4); d_low may be null.
hipError_t launch_interleave(const uint8_t* d_high, const uint8_t* d_low, uint16_t* d_img,
                             size_t npix_total, hipStream_t s);
// pixels -> (px >> 8, px & 0xff) planes for npix_total pixels (multiple of 4, < 2^32).
hipError_t launch_split_planes(const uint16_t* d_img, uint8_t* d_high, uint8_t* d_low,
                               size_t npix_total, hipStream_t s);
// UnextractFrame (cc:850-862) in place on device pixels: value >> shift, byte order of the
// raw file (big_endian != 0: swapped on this little-endian device).
hipError_t launch_unextract(uint16_t* d_px, size_t npix_total, int shift, int big_endian, hipStream_t s);
// Decoded delta image -> byte planes (decoder side, cc:337-338).
hipError_t launch_split_decoded(const Geom& g, const uint16_t* d_img,
                                uint8_t* d_high, uint8_t* d_low, hipStream_t s);

// Forward predictor for nframes frames.  ev, when non-null, points to 5 events
// recorded around [k_stream, k_redo_decide, k_redo_stream, -] (timing pass).
// One workspace serves one stream at a time (its batch counter lives in ws).
hipError_t launch_predict(const Geom& g, const uint16_t* d_frames,
                          uint32_t nframes, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, FwdWorkspace& ws,
                          uint8_t* d_high, uint8_t* d_low, uint8_t* d_preview,
                          uint8_t* d_flags, hipStream_t s, hipEvent_t* ev);

// Inverse predictor.  d_dhigh/d_dlow: delta planes or null.  d_diag, when not
// null, points to 16 dwords the speculative kernels add their counts to: [0] supersteps,
// [1] passes, [2] / [3] maxima, [6] supersteps that went into the long-correction path.
// memo2_hint: take the three-wave kernel's form with the memo behind the second pass (frames
// whose corrections travel: see fpv_inverse.inc) where one is instantiated.
hipError_t launch_inverse(const Geom& g, const uint8_t* d_high,
                          const uint8_t* d_low, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, const uint8_t* d_flags,
                          uint32_t nframes, uint16_t* d_frames, hipStream_t s,
                          uint32_t* d_diag = nullptr, bool memo2_hint = false);

// Overwrites the per-CU "SIMDs running a chain wave" words of the three-wave inverse (current device).
hipError_t set_chain_simd_mask(uint32_t value);

// Name of the kernel launch_inverse picks for a batch of nframes (diagnostics, bench).
const char* inverse_kernel_name(const Geom& g, uint32_t nframes);

}  // namespace fpvhip
#endif  // FPV_KERNELS_H_
