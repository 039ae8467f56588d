// afec_amd/csrc/afx_batch_fetch.cpp -- results of a batch back to the host (the analogue of what the reference appends to
// TFramedScalarData::mValues / TFramedVectorData<W>::mValues, SampleDescriptors.h:152-356), what a batch launches
// (afx_batch_get_info), and its end.  See afx_host.h for the map of the host side.

#include <algorithm>
#include <cstring>

#include "afx_host.h"

using namespace afx::host;

extern "C" {

int afx_batch_fetch_samples(afx_batch* b, int32_t buf, double* dst, int64_t n) {
  if (!b || !dst || buf < 0 || buf >= b->n_bufs || n < 0) return fail(AFX_ERR_INVALID_ARG, "bad argument");
  if (b->pcm_dtype == afx::kPcmF32) return fail(AFX_ERR_INVALID_ARG, "batch does not hold double PCM");
  const int64_t m = std::min<int64_t>(n, b->used[buf]);
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipStreamSynchronize(b->stream));   // the LoadSample write kernel is not waited for at creation
  if (m <= 0) return AFX_OK;
  if (b->pcm_dtype == afx::kPcmF64) {
    HIP_TRY(hipMemcpy(dst, (const double*)b->d_pcm + b->arena_off[buf], (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    return AFX_OK;
  }
  // LoadSample batches keep the float signal and the buffer's FinalScaling: the doubles of TSampleData::mData are their
  // products (SampleAnalyser.cpp:710-718), formed here on demand exactly as the kernels form them
  std::vector<float> tmp((size_t)m);
  HIP_TRY(hipMemcpy(tmp.data(), (const float*)b->d_pcm + b->arena_off[buf], (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
  const double scaling = b->buf_scale[(size_t)buf];
  for (int64_t k = 0; k < m; ++k) dst[k] = (double)tmp[(size_t)k] * scaling;
  return AFX_OK;
}

int64_t afx_batch_total_frames(const afx_batch* batch) { return batch ? batch->total_frames : 0; }

int afx_batch_get_info(const afx_batch* b, afx_batch_info* info) {
  if (!b || !info) return fail(AFX_ERR_INVALID_ARG, "null argument");
  const uint32_t fmask = frames_mask(b->mask);
  info->frame_kernel = b->halfwave ? AFX_FRAME_KERNEL_HALFWAVE : AFX_FRAME_KERNEL_WAVE64;
  info->feature_class = b->halfwave ? afx::frames32_class(fmask) : afx::frames_feature_class(fmask);
  info->pcm_kind = b->pcm_dtype;
  info->chunk_frames = b->chunk_frames;
  info->n_chunks = b->n_chunks;
  info->grid_blocks = b->grid_blocks;
  int64_t samples = 0;
  for (size_t i = 0; i < b->used.size(); ++i) samples += (b->used[i] + 3) & ~(int64_t)3;
  info->arena_bytes = samples * (b->pcm_dtype == afx::kPcmF64 ? 8 : 4);
  return AFX_OK;
}

int afx_batch_fetch(afx_batch* b, afx_out* out) {
  if (!b || !out) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch before afx_batch_run (the pooled workspace would hand back another batch's results)");
  {
    // every requested output must be in the batch mask, whatever the number of frames
    const afx::RecordLayout& lay = b->lay;
    for (const FieldDesc& d : kFields)
      if (out->*(d.out) && lay.*(d.off) < 0) return fail(AFX_ERR_INVALID_ARG, "output requested that is not in the batch mask");
    if (out->magnitude && !(b->mask & AFX_D_MAGNITUDE)) return fail(AFX_ERR_INVALID_ARG, "magnitude not in the batch mask");
    if (out->effective_length && !b->d_efflen && b->n_bufs > 0) return fail(AFX_ERR_INVALID_ARG, "effective_length not in the batch mask");
  }
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (out->frame_offset) std::memcpy(out->frame_offset, b->frame_offset.data(), b->frame_offset.size() * sizeof(int64_t));
  if (out->buf_status) std::memcpy(out->buf_status, b->buf_status.data(), b->buf_status.size() * sizeof(int32_t));
  if (out->effective_length) {
    if (!b->d_efflen) return fail(AFX_ERR_INVALID_ARG, "effective_length not in the batch mask");
    std::vector<int32_t> lt((size_t)b->n_bufs * 6);
    if (!lt.empty()) HIP_TRY(hipMemcpy(lt.data(), b->d_efflen, lt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < b->n_bufs; ++i)
      for (int j = 0; j < 3; ++j) {
        // TAudioMath::SamplesToMs is float arithmetic (AudioMath.inl:134-137); seconds = ms / 1000.0
        // silent leading samples = index of the first sample above the floor (all of them when there is none);
        // the trailing scan stops above that sample (SampleAnalyser.cpp:1731-1746)
        const int64_t first = lt[(size_t)i * 6 + 2 * j], last = lt[(size_t)i * 6 + 2 * j + 1];
        const int64_t lead = (last < 0) ? b->used[i] : first, trail = (last < 0) ? 0 : b->used[i] - 1 - last;
        const int samples = (int)(b->used[i] - lead - trail);
        const float ms = (float)samples / ((float)b->plan->desc.sample_rate / 1000.0f);
        out->effective_length[(size_t)i * 3 + j] = (b->buf_status[i] == AFX_OK && b->used[i] > 0) ? (double)ms / 1000.0 : 0.0;
      }
  }
  const int64_t F = b->total_frames;
  if (F == 0) return AFX_OK;
  const afx::RecordLayout& l = b->lay;
  struct Field { double* dst; int32_t off; int width; };
  std::vector<Field> fields;
  for (const FieldDesc& d : kFields) fields.push_back(Field{out->*(d.out), l.*(d.off), d.width});
  for (const Field& f : fields)
    if (f.dst && f.off < 0) return fail(AFX_ERR_INVALID_ARG, "output requested that is not in the batch mask");
  if (out->magnitude && !(b->mask & AFX_D_MAGNITUDE)) return fail(AFX_ERR_INVALID_ARG, "magnitude not in the batch mask");
  bool any_series = false;
  for (const Field& f : fields) any_series = any_series || f.dst != nullptr;
  if (l.stride > 0 && any_series) {
    std::vector<double> rec((size_t)F * l.stride);
    HIP_TRY(hipMemcpy(rec.data(), b->d_rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (const Field& f : fields) {
      if (!f.dst) continue;
      for (int64_t i = 0; i < F; ++i)
        std::memcpy(f.dst + i * f.width, rec.data() + i * l.stride + f.off, (size_t)f.width * sizeof(double));
    }
  }
  if (out->magnitude)
    HIP_TRY(hipMemcpy(out->magnitude, b->d_mag, (size_t)F * afx::kHalf * sizeof(double), hipMemcpyDeviceToHost));
  return AFX_OK;
}

int afx_batch_fetch_statistics(afx_batch* b, afx_stats_out* out) {
  if (!b || !out) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (b->n_bufs == 0) return AFX_OK;
  if (!b->d_stats) return fail(AFX_ERR_INVALID_ARG, "AFX_D_STATISTICS was not in the batch mask");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_statistics before afx_batch_run");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const afx::RecordLayout& l = b->lay;
  std::vector<double> st((size_t)b->n_bufs * l.stride * 13);
  HIP_TRY(hipMemcpy(st.data(), b->d_stats, st.size() * sizeof(double), hipMemcpyDeviceToHost));
  struct Field { double* dst; int32_t off; int width; };
  std::vector<Field> fields;
  for (const FieldDesc& d : kFields) fields.push_back(Field{out->*(d.stat), l.*(d.off), d.width});
  for (const Field& f : fields) {
    if (!f.dst) continue;
    if (f.off < 0) return fail(AFX_ERR_INVALID_ARG, "statistics requested for a series that is not in the batch mask");
    for (int32_t i = 0; i < b->n_bufs; ++i)
      std::memcpy(f.dst + (size_t)i * f.width * 13, st.data() + ((size_t)i * l.stride + f.off) * 13,
                  (size_t)f.width * 13 * sizeof(double));
  }
  if (out->stats_status)
    for (int32_t i = 0; i < b->n_bufs; ++i) {
      out->stats_status[i] = b->buf_status[i];
    }
  return AFX_OK;
}

int afx_batch_record_layout(const afx_batch* b, int32_t* stride, int32_t* offsets, int32_t* widths) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  if (stride) *stride = b->lay.stride;
  int i = 0;
  for (const FieldDesc& d : kFields) {
    if (offsets) offsets[i] = b->lay.*(d.off);
    if (widths) widths[i] = d.width;
    ++i;
  }
  return AFX_OK;
}

int afx_batch_fetch_records(afx_batch* b, double* records, double* statistics, int64_t* frame_offset, int32_t* buf_status,
                            double* effective_length) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_records before afx_batch_run");
  if (statistics && !b->d_stats && b->n_bufs > 0) return fail(AFX_ERR_INVALID_ARG, "AFX_D_STATISTICS was not in the batch mask");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  const size_t rec_bytes = (size_t)b->total_frames * b->lay.stride * sizeof(double);
  const size_t stat_bytes = (size_t)b->n_bufs * b->lay.stride * 13 * sizeof(double);
  {
    const Download items[2] = {{records, b->d_rec, rec_bytes}, {statistics, b->d_stats, stat_bytes}};
    HIP_TRY(download_through_plan(b, items, 2));
  }
  if (frame_offset) std::memcpy(frame_offset, b->frame_offset.data(), b->frame_offset.size() * sizeof(int64_t));
  if (buf_status) std::memcpy(buf_status, b->buf_status.data(), b->buf_status.size() * sizeof(int32_t));
  if (effective_length) {
    afx_out tmp = {};
    tmp.effective_length = effective_length;
    const int st = afx_batch_fetch(b, &tmp);
    if (st != AFX_OK) return st;
  }
  return AFX_OK;
}

int afx_batch_set_file_info(afx_batch* b, const afx_file_info* info) {
  if (!b || !info) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (!(b->mask & AFX_D_RHYTHM)) return fail(AFX_ERR_INVALID_ARG, "AFX_D_RHYTHM was not in the batch mask");
  if (b->n_bufs > 0) {
    set_rhythm_context(b, info);
    HIP_TRY(hipSetDevice(b->plan->desc.device));
    HIP_TRY(hipMemcpyAsync(b->d_rt_files, b->rt_files.data(), b->rt_files.size() * sizeof(afx::RhythmFile), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));   // rt_files is pageable host memory that may change again
    b->rt_files_dirty = false;
  }
  return AFX_OK;
}

int64_t afx_batch_rhythm_frames(const afx_batch* b, int64_t* offsets) {
  if (!b || b->rt_offset.empty()) {
    if (b && offsets) std::memset(offsets, 0, ((size_t)b->n_bufs + 1) * sizeof(int64_t));
    return 0;
  }
  if (offsets) std::memcpy(offsets, b->rt_offset.data(), b->rt_offset.size() * sizeof(int64_t));
  return b->rt_offset.back();
}

int afx_batch_fetch_rhythm(afx_batch* b, double* onsets, double* scalars, double* onset_statistics) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  if (!(b->mask & AFX_D_RHYTHM)) return fail(AFX_ERR_INVALID_ARG, "AFX_D_RHYTHM was not in the batch mask");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_rhythm before afx_batch_run");
  if (b->n_bufs == 0) return AFX_OK;
  if (onset_statistics && !b->d_rt_stats) return fail(AFX_ERR_INVALID_ARG, "AFX_D_STATISTICS was not in the batch mask");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  const size_t rows = (size_t)b->rt_offset.back();
  {
    const Download items[3] = {{onsets, b->d_rt_onsets, rows * 2 * sizeof(double)},
                               {scalars, b->d_rt_scalars, (size_t)b->n_bufs * AFX_NUM_RHYTHM_SCALARS * sizeof(double)},
                               {onset_statistics, b->d_rt_stats, (size_t)b->n_bufs * 2 * 13 * sizeof(double)}};
    HIP_TRY(download_through_plan(b, items, 3));
  }
  return AFX_OK;
}

int afx_batch_fetch_onset_functions(afx_batch* b, float* odf) {
  if (!b || !odf) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (!(b->mask & AFX_D_RHYTHM)) return fail(AFX_ERR_INVALID_ARG, "AFX_D_RHYTHM was not in the batch mask");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_onset_functions before afx_batch_run");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  const size_t rows = b->rt_offset.empty() ? 0 : (size_t)b->rt_offset.back();
  if (rows) HIP_TRY(hipMemcpyAsync(odf, b->d_rt_odf, rows * 2 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return AFX_OK;
}

void afx_batch_destroy(afx_batch* b) {
  if (!b) return;
  afx_plan* const plan = b->plan;
  struct Release { afx_plan* p; ~Release() { plan_release(p); } } release_plan_last{plan};
  hipSetDevice(b->plan->desc.device);
  if (b->stream) hipStreamSynchronize(b->stream);
  // a run that failed between the rhythm chain's fork and its join leaves kernels on the side stream: they must be
  // done before the workspace goes back to the pool
  if (b->ws && b->ws->side_stream) hipStreamSynchronize(b->ws->side_stream);
#if defined(AFX_STAMPS) && AFX_STAMPS
  stamps_report();
#endif
  ws_release(b->plan, b->ws);
  delete b;
}

}  // extern "C"
