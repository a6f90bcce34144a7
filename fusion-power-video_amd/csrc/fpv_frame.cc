// fpv_frame.cc -- fpvc::Frame (h:75-139): one frame as byte planes with the
// explicit Predict / Compress / Uncompress steps the batch adapters drive.
// Plane arithmetic (split, preview, predictors and their inverses) runs on the
// GPU through single-frame batches of the C ABI; Brotli and the serialisation
// stay on the host exactly like the reference's (cc:643-728, cc:820-846).
#include <brotli/encode.h>
#include <string.h>

#include <future>

#include "fpv_host_internal.h"
#include "fusion_power_video.h"

namespace fpvc {
using namespace internal;

Frame Frame::EMPTY(0, 0);

namespace {
int FrameDevice() { return DeviceList()[0]; }

struct Fwd {  // leased forward context (batch 1, one slot)
  fpvhip_ctx* ctx;
  Fwd(size_t xs, size_t ys, int shift, bool be)
      : ctx(AcquireEncoderCtx(FrameDevice(), (uint32_t)xs, (uint32_t)ys, shift, be, 1, 1)) {
    if (!ctx) Fatal("Frame: fpvhip_create", fpvhip_last_error(nullptr));
  }
  ~Fwd() { ReleaseEncoderCtx(ctx); }
};
struct Inv {  // leased inverse context
  fpvhip_ctx* ctx;
  Inv(size_t xs, size_t ys) : ctx(AcquireDecoderCtx(FrameDevice(), (uint32_t)xs, (uint32_t)ys)) {
    if (!ctx) Fatal("Frame: fpvhip_create_decoder", fpvhip_last_error(nullptr));
  }
  ~Inv() { ReleaseDecoderCtx(ctx); }
};
void Check(int rc, fpvhip_ctx* c, const char* what) {
  if (rc) Fatal(what, fpvhip_last_error(c));
}
// cc:186-214 into a vector (Frame::Uncompress does not check sizes)
void BrotliToVector(std::vector<uint8_t>* v) {
  std::vector<uint8_t> out;
  size_t cap = v->size() * 8 + 4096;
  for (;;) {
    out.resize(cap);
    size_t pos = 0, got = 0;
    const bool ok = BrotliInto(v->data(), v->size(), &pos, out.data(), cap, &got);
    if (ok || got < cap) {  // done, or a corrupt stream: keep what was produced
      out.resize(got);
      break;
    }
    cap *= 4;
  }
  v->swap(out);
}
}  // namespace

size_t Frame::MaxCompressedPlaneSize(size_t xsize, size_t ysize) {
  return BrotliEncoderMaxCompressedSize(xsize * ysize);
}
size_t Frame::MaxCompressedPreviewSize(size_t xsize, size_t ysize) {
  return BrotliEncoderMaxCompressedSize(xsize * ysize / 16);
}
size_t Frame::MaxCompressedPlaneSize() { return BrotliEncoderMaxCompressedSize(size_); }
size_t Frame::MaxCompressedPreviewSize() { return BrotliEncoderMaxCompressedSize(size_ / 16); }

// cc:370-451: the split runs on the GPU right away (flags() must be correct
// after construction); the pixels are kept so that Predict is ONE fused pass.
Frame::Frame(size_t xsize, size_t ysize, const uint16_t* image, int shift_to_left_align,
             bool big_endian, int64_t timestamp)
    : xsize_(xsize), ysize_(ysize), size_(xsize * ysize), timestamp_(timestamp),
      shift_(shift_to_left_align), big_endian_(big_endian) {
  if (!image) return;
  state_ = FrameState::RAW;
  Fwd f(xsize_, ysize_, shift_, big_endian_);
  high_.resize(size_);
  if (shift_ != 8) low_.resize(size_);
  int no_low = 0;
  Check(fpvhip_split_frame(f.ctx, image, high_.data(), shift_ != 8 ? low_.data() : nullptr, &no_low),
        f.ctx, "fpvhip_split_frame");
  if (no_low) flags_ |= FrameFlags::NO_LOW_BYTES;
  pixels_ = std::make_shared<std::vector<uint16_t>>(image, image + size_);
}

// cc:453-465
Frame::Frame(size_t xsize, size_t ysize, const uint8_t* image, int64_t timestamp)
    : xsize_(xsize), ysize_(ysize), size_(xsize * ysize), flags_(FrameFlags::NO_LOW_BYTES),
      timestamp_(timestamp) {
  if (image) {
    state_ = FrameState::RAW;
    high_.assign(image, image + size_);
  }
}

// cc:467-489, including its two no-op assignments to the parameters (cc:482, 485)
Frame::Frame(size_t xsize, size_t ysize, uint8_t flags, uint8_t state, std::vector<uint8_t>&& high,
             std::vector<uint8_t>&& low, std::vector<uint8_t>&& preview, int64_t timestamp)
    : preview_(std::move(preview)), high_(std::move(high)), low_(std::move(low)), xsize_(xsize),
      ysize_(ysize), size_(xsize * ysize), flags_(flags), state_(state), timestamp_(timestamp) {
  if (preview_.empty()) state_ &= ~FrameState::PREVIEW_GENERATED;
  if (!low_.empty()) flags_ &= ~FrameFlags::NO_LOW_BYTES;
}

// cc:777-785
void Frame::Predict(Frame& delta_frame) {
  const uint8_t predicted = FrameState::DELTA_PREDICTED | FrameState::CG_PREDICTED;
  if (state_ & FrameState::CG_PREDICTED) {
    const bool want_delta = delta_frame.state() > FrameState::EMPTY;
    if (!want_delta || (state_ & FrameState::DELTA_PREDICTED)) return;  // nothing left to do
  }
  if ((state_ & (predicted | FrameState::COMPRESSED)) || high_.size() != size_ || size_ == 0)
    Fatal("Frame::Predict", "only frames in the RAW state can be predicted on the GPU");
  const bool with_delta = delta_frame.state() > FrameState::EMPTY;
  const uint8_t *gh, *gl, *gp, *gf;
  const size_t npv = (xsize_ / 4) * (ysize_ / 4);
  auto run = [&](fpvhip_ctx* ctx, bool planes) {
    if (with_delta)
      Check(fpvhip_set_delta_planes(ctx, delta_frame.high_.data(),
                                    delta_frame.low_.size() == size_ ? delta_frame.low_.data() : nullptr),
            ctx, "fpvhip_set_delta_planes");
    if (planes) {
      memcpy(fpvhip_stage_plane(ctx, 0, 0, 0), high_.data(), size_);
      if (low_.size() == size_) memcpy(fpvhip_stage_plane(ctx, 0, 0, 1), low_.data(), size_);
      else memset(fpvhip_stage_plane(ctx, 0, 0, 1), 0, size_);
      Check(fpvhip_predict_planes_submit(ctx, 0, 1, with_delta), ctx, "fpvhip_predict_planes_submit");
    } else {
      memcpy(fpvhip_stage_in(ctx, 0, 0), pixels_->data(), size_ * 2);
      Check(fpvhip_predict_submit(ctx, 0, 1, with_delta), ctx, "fpvhip_predict_submit");
    }
    Check(fpvhip_predict_wait(ctx, 0, &gh, &gl, &gp, &gf), ctx, "fpvhip_predict_wait");
    memcpy(high_.data(), gh, size_);
    if (low_.size() == size_) memcpy(low_.data(), gl, size_);
    preview_.assign(gp, gp + npv);
    flags_ = (uint8_t)((flags_ & FrameFlags::NO_LOW_BYTES) |
                       (gf[0] & (FrameFlags::USE_DELTA | FrameFlags::USE_CG)));
  };
  if (pixels_) {
    Fwd f(xsize_, ysize_, shift_, big_endian_);
    run(f.ctx, false);
  } else {
    Fwd f(xsize_, ysize_, 0, false);
    run(f.ctx, true);
  }
  pixels_.reset();
  state_ &= ~FrameState::RAW;
  state_ |= FrameState::PREVIEW_GENERATED | FrameState::CG_PREDICTED;
  if (with_delta) state_ |= FrameState::DELTA_PREDICTED;
}

// cc:738-745 + cc:643-688
void Frame::Compress(Frame& delta_frame) {
  if (state_ & FrameState::COMPRESSED) return;
  Predict(delta_frame);
  std::vector<uint8_t> out;
  if (!BrotliAppend(high_.data(), size_, &out)) Fatal("Brotli", "high plane");
  high_.swap(out);
  if (flags_ & FrameFlags::NO_LOW_BYTES) {
    low_.clear();
  } else {
    out.clear();
    if (!BrotliAppend(low_.data(), size_, &out)) Fatal("Brotli", "low plane");
    low_.swap(out);
  }
  if (state_ & FrameState::PREVIEW_GENERATED) {
    out.clear();
    if (!BrotliAppend(preview_.data(), preview_.size(), &out)) Fatal("Brotli", "preview");
    preview_.swap(out);
  }
  state_ &= ~FrameState::RAW;
  state_ |= FrameState::COMPRESSED;
}

// cc:787-818 and cc:690-728
void Frame::CompressPredicted(size_t* encoded_high_size, uint8_t* encoded_high_buffer,
                              size_t* encoded_low_size, uint8_t* encoded_low_buffer,
                              size_t* encoded_preview_size, uint8_t* encoded_preview_buffer,
                              bool parallel) {
  auto copy_out = [](const std::vector<uint8_t>& v, size_t* sz, uint8_t* buf) {
    if (buf && *sz >= v.size()) {
      memcpy(buf, v.data(), v.size());
      *sz = v.size();
    } else {
      *sz = 0;
    }
  };
  if (state_ & FrameState::COMPRESSED) {
    copy_out(high_, encoded_high_size, encoded_high_buffer);
    copy_out(low_, encoded_low_size, encoded_low_buffer);
    copy_out(preview_, encoded_preview_size, encoded_preview_buffer);
    return;
  }
  auto brotli = [](const uint8_t* in, size_t n, size_t* sz, uint8_t* buf) {
    BrotliEncoderCompress(1, BROTLI_DEFAULT_WINDOW, BROTLI_DEFAULT_MODE, n, in, sz, buf);
  };
  std::future<void> low_task;
  if (!encoded_low_buffer || (flags_ & FrameFlags::NO_LOW_BYTES)) {
    *encoded_low_size = 0;
  } else if (parallel) {  // the low plane is the expensive one: overlap it (cc:697-704)
    low_task = std::async(std::launch::async, [&] { brotli(low_.data(), size_, encoded_low_size, encoded_low_buffer); });
  } else {
    brotli(low_.data(), size_, encoded_low_size, encoded_low_buffer);
  }
  if (encoded_high_buffer) brotli(high_.data(), size_, encoded_high_size, encoded_high_buffer);
  else *encoded_high_size = 0;
  if (encoded_preview_buffer && (state_ & FrameState::PREVIEW_GENERATED))
    brotli(preview_.data(), preview_.size(), encoded_preview_size, encoded_preview_buffer);
  else *encoded_preview_size = 0;
  if (low_task.valid()) low_task.wait();
}

// cc:747-775 with the two Optionally-Unapply steps (cc:612-641, cc:595-610)
void Frame::Uncompress(Frame& delta_frame) {
  if (state_ & FrameState::COMPRESSED) {
    if (!high_.empty()) BrotliToVector(&high_);
    if (!(low_.empty() || (flags_ & FrameFlags::NO_LOW_BYTES))) BrotliToVector(&low_);
    if ((state_ & FrameState::PREVIEW_GENERATED) && !preview_.empty()) BrotliToVector(&preview_);
    state_ &= ~FrameState::COMPRESSED;
  }
  const bool undo_cg = (state_ & FrameState::CG_PREDICTED) && (flags_ & FrameFlags::USE_CG);
  const bool undo_delta = (state_ & FrameState::DELTA_PREDICTED) && (flags_ & FrameFlags::USE_DELTA) &&
                          delta_frame.state() != FrameState::EMPTY;
  const bool planes_ok = high_.size() == size_ && size_ > 0;
  if (planes_ok && (undo_cg || undo_delta)) {
    Inv v(xsize_, ysize_);
    if (undo_delta)
      Check(fpvhip_set_delta_planes(v.ctx, delta_frame.high_.data(),
                                    delta_frame.low_.size() == size_ ? delta_frame.low_.data() : nullptr),
            v.ctx, "fpvhip_set_delta_planes");
    const bool has_low = low_.size() == size_;
    memcpy(fpvhip_stage_plane(v.ctx, 0, 0, 0), high_.data(), size_);
    if (has_low) memcpy(fpvhip_stage_plane(v.ctx, 0, 0, 1), low_.data(), size_);
    const uint8_t f = (uint8_t)((undo_delta ? 1 : 0) | (undo_cg ? 2 : 0) | (has_low ? 0 : 4));
    Check(fpvhip_inverse_planes_submit(v.ctx, 0, 1, &f), v.ctx, "fpvhip_inverse_planes_submit");
    const uint8_t *gh, *gl;
    Check(fpvhip_predict_wait(v.ctx, 0, &gh, &gl, nullptr, nullptr), v.ctx, "fpvhip_predict_wait");
    memcpy(high_.data(), gh, size_);
    if (has_low) memcpy(low_.data(), gl, size_);
  }
  if (undo_cg) {
    const size_t pw = xsize_ / 4, ph = ysize_ / 4;
    if ((state_ & FrameState::PREVIEW_GENERATED) && preview_.size() == size_ / 16 && pw && ph &&
        pw * ph == preview_.size()) {
      Inv v(pw, ph);
      memcpy(fpvhip_stage_plane(v.ctx, 0, 0, 0), preview_.data(), preview_.size());
      const uint8_t f = 2 | 4;
      if ((pw * ph) % 4 == 0) {
        Check(fpvhip_inverse_planes_submit(v.ctx, 0, 1, &f), v.ctx, "fpvhip_inverse_planes_submit");
        const uint8_t* gh;
        Check(fpvhip_predict_wait(v.ctx, 0, &gh, nullptr, nullptr, nullptr), v.ctx, "fpvhip_predict_wait");
        memcpy(preview_.data(), gh, preview_.size());
      } else {  // odd preview size: pixels out, keep the high bytes
        Check(fpvhip_inverse_submit(v.ctx, 0, 1, &f), v.ctx, "fpvhip_inverse_submit");
        const uint16_t* px;
        Check(fpvhip_inverse_wait(v.ctx, 0, &px), v.ctx, "fpvhip_inverse_wait");
        for (size_t i = 0; i < preview_.size(); i++) preview_[i] = (uint8_t)(px[i] >> 8);
      }
    }
    flags_ &= ~FrameFlags::USE_CG;
    state_ &= ~FrameState::CG_PREDICTED;
    if (state_ < FrameState::DELTA_PREDICTED) state_ |= FrameState::RAW;
  }
  if (undo_delta) {
    flags_ &= ~FrameFlags::USE_DELTA;
    state_ &= ~FrameState::DELTA_PREDICTED;
    if (state_ < FrameState::DELTA_PREDICTED) state_ |= FrameState::RAW;
  }
}

// cc:820-828
void Frame::OutputCore(std::vector<uint8_t>* out) {
  if (!(state_ & FrameState::COMPRESSED)) return;
  out->reserve(out->size() + 1 + high_.size() + low_.size());
  out->push_back(flags_);
  out->insert(out->end(), low_.begin(), low_.end());
  out->insert(out->end(), high_.begin(), high_.end());
}

// cc:830-846
void Frame::OutputFull(std::vector<uint8_t>* out) {
  if (!(state_ & FrameState::COMPRESSED)) return;
  const size_t total = (9 + 1 + preview_.size()) + (1 + high_.size() + low_.size());
  out->reserve(out->size() + total);
  Append32((uint32_t)total, out);
  out->push_back(0);
  Append32((uint32_t)(preview_.size() + 1), out);
  out->push_back((uint8_t)((flags_ & FrameFlags::USE_CG) | FrameFlags::NO_LOW_BYTES));
  out->insert(out->end(), preview_.begin(), preview_.end());
  OutputCore(out);
}

}  // namespace fpvc
