// afec_amd/host/Crawler.cpp -- see Crawler.h.
#include "Crawler.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <time.h>

#include <sched.h>
#include <sys/stat.h>

#include "../../include/afx.h"
#include "SqlitePool.h"
#include "WaveFile.h"

namespace afec {

namespace {

double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

double ProcessCpuSeconds() {
  timespec t;
  ::clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

// 64-bit FNV-1a over 8-byte words (row digests: equality of results, not cryptography)
struct TDigest {
  uint64_t mHash = 1469598103934665603ull;
  void Add(const void* p, size_t Bytes) {
    const unsigned char* b = (const unsigned char*)p;
    size_t i = 0;
    for (; i + 8 <= Bytes; i += 8) {
      uint64_t w;
      std::memcpy(&w, b + i, 8);
      mHash = (mHash ^ w) * 1099511628211ull;
    }
    for (; i < Bytes; ++i) mHash = (mHash ^ b[i]) * 1099511628211ull;
  }
};

double ThreadCpuSeconds() {
  timespec t;
  ::clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

// page-locked host memory (afx_host_alloc), grown on demand
struct TPinned {
  void* mp = nullptr;
  size_t mBytes = 0;
  ~TPinned() { afx_host_free(mp); }
  void Reserve(size_t Bytes) {
    if (Bytes <= mBytes) return;
    afx_host_free(mp);
    mBytes = Bytes + Bytes / 4;
    mp = afx_host_alloc((int64_t)mBytes);
    if (!mp) { mBytes = 0; throw TReadableException("page-locked host memory exhausted"); }
  }
  // the same, keeping the first Used bytes
  void Grow(size_t Bytes, size_t Used) {
    if (Bytes <= mBytes) return;
    const size_t NewBytes = Bytes + Bytes / 4;
    void* pNew = afx_host_alloc((int64_t)NewBytes);
    if (!pNew) throw TReadableException("page-locked host memory exhausted");
    if (Used) std::memcpy(pNew, mp, Used);
    afx_host_free(mp);
    mp = pNew;
    mBytes = NewBytes;
  }
};

// page-locking memory costs milliseconds per allocation: result buffers are recycled between the workers (which fill
// them) and the writer (which hands them back)
class TPinnedPool {
public:
  std::unique_ptr<TPinned> Acquire(size_t Bytes) {
    std::unique_ptr<TPinned> p;
    {
      std::lock_guard<std::mutex> Lock(mMutex);
      // the smallest free buffer that is large enough, else the largest one (it is grown)
      size_t Best = mFree.size();
      for (size_t i = 0; i < mFree.size(); ++i) {
        const bool Fits = mFree[i]->mBytes >= Bytes;
        if (Best == mFree.size()) Best = i;
        else {
          const bool BestFits = mFree[Best]->mBytes >= Bytes;
          if ((Fits && (!BestFits || mFree[i]->mBytes < mFree[Best]->mBytes)) || (!Fits && !BestFits && mFree[i]->mBytes > mFree[Best]->mBytes)) Best = i;
        }
      }
      if (Best < mFree.size()) {
        p = std::move(mFree[Best]);
        mFree.erase(mFree.begin() + (long)Best);
      }
    }
    if (!p) p.reset(new TPinned);
    p->Reserve(Bytes);
    return p;
  }
  void Release(std::unique_ptr<TPinned> p) {
    if (!p) return;
    std::lock_guard<std::mutex> Lock(mMutex);
    mFree.push_back(std::move(p));
  }

private:
  std::mutex mMutex;
  std::vector<std::unique_ptr<TPinned>> mFree;
};

// what a worker hands to the writer: one analysed batch (its result buffers travel with it)
// The column values of a batch's files (461 per file, ~68 KB of msgpack for a one-second file), built by the worker that
// analysed the batch so that the one writer thread only binds and steps; recycled between batches: names and BLOB
// capacities persist (RefillLowLevelColumns)
struct TRowSet {
  std::vector<std::vector<TColumn>> mRows;
};
class TRowSetPool {
public:
  std::unique_ptr<TRowSet> Acquire() {
    std::lock_guard<std::mutex> Lock(mMutex);
    if (mFree.empty()) return std::unique_ptr<TRowSet>(new TRowSet);
    std::unique_ptr<TRowSet> p = std::move(mFree.back());
    mFree.pop_back();
    return p;
  }
  void Release(std::unique_ptr<TRowSet> p) {
    if (!p) return;
    std::lock_guard<std::mutex> Lock(mMutex);
    if (mFree.size() < 32) mFree.push_back(std::move(p));
  }

private:
  std::mutex mMutex;
  std::vector<std::unique_ptr<TRowSet>> mFree;
};

struct TFinishedBatch {
  std::vector<const TCrawlFile*> mFiles;
  std::vector<TFileProperties> mProperties;
  std::vector<std::string> mFailed;          // non-empty: the file is a failed sample with this reason
  std::vector<char> mSkipped;                // 1: not analysed and not recorded (sampling rate other than the analyser's)
  std::vector<int> mBatchIndex;              // file -> index inside mResults, -1 for files that never reached the GPU
  TRecordBatch mResults;
  std::unique_ptr<TPinned> mpRecords, mpStatistics, mpRhythm;
  std::unique_ptr<TRowSet> mpRows;           // with a database: mRows[k] = the column values of batch file k
};

class TBoundedQueue {
public:
  explicit TBoundedQueue(size_t Capacity) : mCapacity(Capacity) {}
  void Push(std::unique_ptr<TFinishedBatch> p) {
    std::unique_lock<std::mutex> Lock(mMutex);
    mNotFull.wait(Lock, [&] { return mItems.size() < mCapacity; });
    mItems.push_back(std::move(p));
    mNotEmpty.notify_one();
  }
  // nullptr once Close() was called and the queue has drained
  std::unique_ptr<TFinishedBatch> Pop() {
    std::unique_lock<std::mutex> Lock(mMutex);
    mNotEmpty.wait(Lock, [&] { return !mItems.empty() || mClosed; });
    if (mItems.empty()) return nullptr;
    std::unique_ptr<TFinishedBatch> p = std::move(mItems.front());
    mItems.pop_front();
    mNotFull.notify_one();
    return p;
  }
  void Close() {
    std::lock_guard<std::mutex> Lock(mMutex);
    mClosed = true;
    mNotEmpty.notify_all();
  }

private:
  std::mutex mMutex;
  std::condition_variable mNotFull, mNotEmpty;
  std::deque<std::unique_ptr<TFinishedBatch>> mItems;
  size_t mCapacity;
  bool mClosed = false;
};

}  // namespace

double UsableHostCpus() {
  double Cpus = 0;
  cpu_set_t Set;
  if (::sched_getaffinity(0, sizeof(Set), &Set) == 0) Cpus = (double)CPU_COUNT(&Set);
  if (Cpus < 1) Cpus = (double)std::thread::hardware_concurrency();
  if (Cpus < 1) Cpus = 1;
  // the container's CPU bandwidth limit (the GPU pool shows 256 hardware threads and allows 16 CPUs)
  double Quota = 0;
  if (std::FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
    char Text[64] = {0};
    double Period = 0;
    if (std::fscanf(f, "%63s %lf", Text, &Period) == 2 && std::strcmp(Text, "max") != 0 && Period > 0) Quota = std::atof(Text) / Period;
    std::fclose(f);
  } else {
    double Q = 0, P = 0;                                                      // cgroup v1
    if (std::FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(q, "%lf", &Q) != 1) Q = 0; std::fclose(q); }
    if (std::FILE* p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(p, "%lf", &P) != 1) P = 0; std::fclose(p); }
    if (Q > 0 && P > 0) Quota = Q / P;
  }
  return (Quota > 0 && Quota < Cpus) ? Quota : Cpus;
}

int WorkersPerDeviceFor(int NumberOfDevices) {
  const int PerDevice = (int)(UsableHostCpus() / (double)(NumberOfDevices < 1 ? 1 : NumberOfDevices));
  return PerDevice < 1 ? 1 : (PerDevice > 5 ? 5 : PerDevice);
}

// What outlives one crawl: the analysers (one plan per device, with its pooled device workspaces) and the page-locked
// buffers.  Setting these up costs ~65 ms on an MI355X box -- as much as analysing 8 000 one-second files.
struct TCrawler::TImpl {
  std::vector<int> mDevices;
  int mSampleRate, mFftFrameSize, mHopFrameSize;
  std::vector<std::unique_ptr<TSampleAnalyser>> mAnalysers;
  TPinnedPool mRecordPool, mStatisticsPool, mRhythmPool, mStagingPool;   // one pool per kind of buffer: nothing regrows
  TRowSetPool mRowPool;
  int mHardwareQueuesInEnvironment = 0;
};

TCrawler::TCrawler(const TCrawlOptions& Options) : mpImpl(new TImpl) {
  mpImpl->mDevices = Options.mDevices;
  mpImpl->mSampleRate = Options.mSampleRate; mpImpl->mFftFrameSize = Options.mFftFrameSize; mpImpl->mHopFrameSize = Options.mHopFrameSize;
  if (Options.mDevices.empty()) { delete mpImpl; throw TReadableException("CrawlWaveFiles: no device given"); }
  try {
    // one analyser (plan) per device, shared by that device's workers like the reference's const analyser; the
    // workers' waits for the device sleep instead of spinning (eight spinning threads per GPU would need eight CPUs
    // per GPU for the same throughput); the hardware-queue wish has to reach the runtime before its first call
    // (setenv is not safe against a concurrent getenv in another thread and has no effect once the HIP runtime is up:
    // an embedding application sets the variable itself at process start -- bench.py does -- and passes
    // mHardwareQueues = 0; what the environment says when the crawler is built is reported in
    // TCrawlStatistics::mHardwareQueuesInEnvironment)
    if (Options.mHardwareQueues > 0)
      ::setenv("GPU_MAX_HW_QUEUES", std::to_string(Options.mHardwareQueues).c_str(), /*overwrite=*/0);
    {
      const char* const q = std::getenv("GPU_MAX_HW_QUEUES");
      mpImpl->mHardwareQueuesInEnvironment = q ? std::atoi(q) : 0;
    }
    for (int Device : Options.mDevices) {
      mpImpl->mAnalysers.emplace_back(new TSampleAnalyser(Options.mSampleRate, Options.mFftFrameSize, Options.mHopFrameSize, Device,
                                                          Options.mFrameKernel));
      mpImpl->mAnalysers.back()->SetSleepingWaits(Options.mSleepingWaits);
    }
  } catch (...) {
    delete mpImpl;
    throw;
  }
}

TCrawler::~TCrawler() { delete mpImpl; }

TCrawlStatistics CrawlWaveFiles(const std::vector<TCrawlFile>& Files, const TCrawlOptions& Options) {
  TCrawler Crawler(Options);
  return Crawler.Crawl(Files, Options);
}

TCrawlStatistics TCrawler::Crawl(const std::vector<TCrawlFile>& Files, const TCrawlOptions& Options) {
  if (Options.mDevices != mpImpl->mDevices || Options.mSampleRate != mpImpl->mSampleRate ||
      Options.mFftFrameSize != mpImpl->mFftFrameSize || Options.mHopFrameSize != mpImpl->mHopFrameSize)
    throw TReadableException("TCrawler::Crawl: devices / geometry differ from the crawler's");
  const int G = (int)Options.mDevices.size();
  const int W = Options.mWorkersPerDevice < 1 ? WorkersPerDeviceFor(G) : Options.mWorkersPerDevice;
  const int FilesPerBatch = Options.mFilesPerBatch < 1 ? 1 : Options.mFilesPerBatch;
  std::vector<std::unique_ptr<TSampleAnalyser>>& Analysers = mpImpl->mAnalysers;
  TPinnedPool& Pool = mpImpl->mRecordPool;
  TPinnedPool& StatisticsPool = mpImpl->mStatisticsPool;
  TPinnedPool& RhythmPool = mpImpl->mRhythmPool;
  TPinnedPool& StagingPool = mpImpl->mStagingPool;
  TRowSetPool& RowPool = mpImpl->mRowPool;
  std::unique_ptr<TSqliteSampleDescriptorPool> pPool;
  if (!Options.mDatabasePath.empty()) pPool.reset(new TSqliteSampleDescriptorPool(Options.mDatabasePath, Options.mDatabasePragmas));

  // shards: file i -> device i mod G, in crawl order
  std::vector<std::vector<const TCrawlFile*>> Shard((size_t)G);
  for (size_t i = 0; i < Files.size(); ++i) Shard[(size_t)ShardOfFile((int64_t)i, G)].push_back(&Files[i]);
  std::vector<size_t> Cursor((size_t)G, 0);
  std::vector<std::mutex> CursorMutex((size_t)G);
  const int64_t BytesPerBatch = Options.mBytesPerBatch < 1 ? 1 : Options.mBytesPerBatch;
  auto FileBytes = [](const TCrawlFile& f) -> int64_t {
    if (f.mpImage) return (int64_t)f.mImageSize;
    struct stat St;
    return ::stat(f.mFileName.c_str(), &St) == 0 ? (int64_t)St.st_size : 0;   // a file that is not there fails when it is opened
  };

  TCrawlStatistics Total;
  Total.mFilesPerDevice.assign((size_t)G, 0);
  Total.mPcmBytesPerDevice.assign((size_t)G, 0);
  Total.mSecondsPerDevice.assign((size_t)G, 0.0);
  Total.mWorkersPerDevice = W;
  Total.mUsableHostCpus = UsableHostCpus();
  if (Options.mRowDigests) Total.mRowDigests.assign(Files.size(), 0);
  const double Start = Now();
  Total.mHardwareQueuesInEnvironment = mpImpl->mHardwareQueuesInEnvironment;
  double PhaseSeconds[2] = {0, 0};   // summed over workers: parse + staging copy, GPU round trip
  double GpuSeconds[3] = {0, 0, 0};  // of the round trip: upload + LoadSample, kernels enqueue, download + wait
  double PhaseCpuSeconds[3] = {0, 0, 0};   // CPU time of the threads: workers parse + staging, workers GPU round trip, writer
  std::mutex StatMutex;
  TBoundedQueue Queue((size_t)(2 * G * W));
  std::atomic<bool> Abort(false);      // an error ends the crawl: nothing more is analysed or written
  std::atomic<bool> Stopped(false);    // TCrawlOptions::mpAbortRequested: no new batches; what was analysed is written
  std::string FirstError;

  // ---- one batch on the GPU, with the reference's failure semantics (SampleAnalyser.cpp:368-408: a file that cannot
  // be analysed gets a failed row and the crawl goes on) for errors of the device path: a batch whose GPU round trip
  // fails -- out of device memory, results that do not fit, a failed runtime call -- is cut in halves and each half is
  // tried on its own; a single file is tried twice and then recorded as "Sample failed to analyse: ...".  Only a device
  // that no longer answers ends the crawl.
  std::atomic<int> NextOrdinal(0);
  std::atomic<int> FaultBudget(Options.mTestFailAttempts < 0 ? (1 << 30) : Options.mTestFailAttempts);
  const int64_t DeviceBytesPerBatch = Options.mDeviceBytesPerBatch < 1 ? 1 : Options.mDeviceBytesPerBatch;
  // device memory a file needs once it is analysed: its converted samples as floats, the raw upload, 8 KiB of
  // magnitudes + ~1 KiB of records per 1024-sample hop, the rhythm tracker's rows (a converted file may be far larger
  // than its bytes on disk: a header that claims a low sampling rate)
  auto DeviceBytesOf = [&](const TDecodedSample& s) -> int64_t {
    const int64_t Converted = TSampleAnalyser::ConvertedSampleFrames(s, Options.mSampleRate);
    return Converted * 16 + s.mNumberOfSampleFrames * s.mNumberOfChannels * 4;
  };

  struct TWork {
    std::unique_ptr<TFinishedBatch> mpDone;
    std::vector<TDecodedSample> mDecoded;
    int mOrdinal = 0;
  };
  // the two halves of a batch: decoded files [0, m) and [m, K); files that never reached the GPU stay with the first
  auto Split = [](TWork& Whole, TWork& A, TWork& B) {
    const size_t K = Whole.mDecoded.size(), m = K / 2;
    A.mpDone.reset(new TFinishedBatch); B.mpDone.reset(new TFinishedBatch);
    A.mOrdinal = B.mOrdinal = Whole.mOrdinal;
    A.mDecoded.assign(Whole.mDecoded.begin(), Whole.mDecoded.begin() + (long)m);
    B.mDecoded.assign(Whole.mDecoded.begin() + (long)m, Whole.mDecoded.end());
    TFinishedBatch& W = *Whole.mpDone;
    for (size_t i = 0; i < W.mFiles.size(); ++i) {
      const int k = W.mBatchIndex[i];
      TFinishedBatch& T = (k >= (int)m) ? *B.mpDone : *A.mpDone;
      T.mFiles.push_back(W.mFiles[i]);
      T.mProperties.push_back(W.mProperties[i]);
      T.mFailed.push_back(W.mFailed[i]);
      T.mSkipped.push_back(W.mSkipped[i]);
      T.mBatchIndex.push_back(k < 0 ? -1 : (k >= (int)m ? k - (int)m : k));
    }
  };

  std::function<void(int, TWork&, int)> Process = [&](int d, TWork& Work, int Attempt) {
    const TSampleAnalyser& Analyser = *Analysers[(size_t)d];
    TFinishedBatch& Done = *Work.mpDone;
    const std::vector<TDecodedSample>& Decoded = Work.mDecoded;
    const size_t n = Done.mFiles.size();
    // a batch whose converted samples would not fit the device budget is cut before it is tried
    if (Decoded.size() > 1) {
      int64_t Need = 0;
      for (const TDecodedSample& s : Decoded) Need += DeviceBytesOf(s);
      if (Need > DeviceBytesPerBatch) {
        TWork A, B;
        Split(Work, A, B);
        Process(d, A, 0);
        Process(d, B, 0);
        return;
      }
    }
    const double tGpu0 = Now(), cGpu0 = ThreadCpuSeconds();
    int64_t Frames = 0, ResultBytes = 0, PcmBytes = 0;
    for (const TDecodedSample& s : Decoded) PcmBytes += s.mNumberOfSampleFrames * s.mNumberOfChannels * (s.mFormat == AFX_RAW_I16 ? 2 : (s.mFormat == AFX_RAW_I24 ? 3 : (s.mFormat == AFX_RAW_F64 ? 8 : 4)));
    if (!Decoded.empty()) {
      try {
        if (Work.mOrdinal == Options.mTestFailBatch && FaultBudget.fetch_sub(1) > 0)
          throw TReadableException("GPU feature extraction failed: injected fault (TCrawlOptions::mTestFailBatch)");
        Done.mpStatistics = StatisticsPool.Acquire(Decoded.size() * (size_t)TSampleAnalyser::kMaxStride * 13 * sizeof(double));
        Done.mpRhythm = RhythmPool.Acquire(Analyser.RhythmDoubles(Decoded) * sizeof(double));
        // frames are at most samples / hop + 2 per file (LoadSample pads by up to a frame)
        size_t MaxFrames = 0;
        for (const TDecodedSample& s : Decoded) MaxFrames += (size_t)(TSampleAnalyser::ConvertedSampleFrames(s, Options.mSampleRate) / Options.mHopFrameSize) + 3;
        size_t Capacity = MaxFrames * (size_t)TSampleAnalyser::kMaxStride;
        int Attempts = 0;
        for (;;) {
          if (Done.mpRecords) Done.mpRecords->Reserve(Capacity * sizeof(double));
          else Done.mpRecords = Pool.Acquire(Capacity * sizeof(double));
          if (Analyser.AnalyzeToRecords(Decoded, (double*)Done.mpRecords->mp, Done.mpRecords->mBytes / sizeof(double),
                                        (double*)Done.mpStatistics->mp, (double*)Done.mpRhythm->mp,
                                        Done.mpRhythm->mBytes / sizeof(double), Done.mResults))
            break;
          if (++Attempts > 6) throw TReadableException("AnalyzeToRecords: the results do not fit the largest buffers tried");
          Capacity *= 2;
          Done.mpRhythm->Reserve(2 * Done.mpRhythm->mBytes);
        }
      } catch (const TReadableException& e) {
        Pool.Release(std::move(Done.mpRecords));
        StatisticsPool.Release(std::move(Done.mpStatistics));
        RhythmPool.Release(std::move(Done.mpRhythm));
        Done.mResults = TRecordBatch();
        if (Options.mTestDeviceLost || !Analyser.DeviceUsable()) throw;   // nothing more can be analysed: the crawl ends
        {
          std::lock_guard<std::mutex> Lock(StatMutex);
          Total.mRetriedBatches += 1;
        }
        if (Decoded.size() > 1) {
          TWork A, B;
          Split(Work, A, B);
          Process(d, A, 0);
          Process(d, B, 0);
          return;
        }
        if (Attempt == 0) { Process(d, Work, 1); return; }
        // one file, twice, on a device that still answers: the file's row says so (SampleAnalyser.cpp:397-408)
        for (size_t i = 0; i < n; ++i)
          if (Done.mBatchIndex[i] >= 0) {
            Done.mFailed[i] = std::string("Sample failed to analyse: ") + e.what();
            Done.mBatchIndex[i] = -1;
          }
        Work.mDecoded.clear();
        std::lock_guard<std::mutex> Lock(StatMutex);
        Total.mDeviceFailedFiles += 1;
      }
    }
    if (!Work.mDecoded.empty()) {
      Frames = Done.mResults.mFrameOffset.back();
      ResultBytes = (Frames * Done.mResults.mStride + (int64_t)Decoded.size() * Done.mResults.mStride * 13 +
                     Done.mResults.mRhythmOffset.back() * 2 + (int64_t)Decoded.size() * 40) * 8;
      for (size_t i = 0; i < n; ++i) {
        const int k = Done.mBatchIndex[i];
        // the per-file status of the LoadSample front end (a buffer it cannot take): a load failure (SampleAnalyser.cpp:372-387)
        if (k >= 0 && Done.mResults.mStatus[(size_t)k] != AFX_OK)
          Done.mFailed[i] = std::string("Sample failed to load: ") + afx_status_str(Done.mResults.mStatus[(size_t)k]);
      }
      // a digest of everything the device returned for a file (TCrawlOptions::mRowDigests): the file's slot is this
      // worker's alone, no lock
      if (Options.mRowDigests) {
        const TRecordBatch& R = Done.mResults;
        for (size_t i = 0; i < n; ++i) {
          const int k = Done.mBatchIndex[i];
          if (k < 0 || !Done.mFailed[i].empty()) continue;
          TDigest D;
          const int64_t f0 = R.mFrameOffset[(size_t)k], f1 = R.mFrameOffset[(size_t)k + 1];
          D.Add(R.mpRecords + f0 * R.mStride, (size_t)((f1 - f0) * R.mStride) * sizeof(double));
          D.Add(R.mpStatistics + (size_t)k * (size_t)R.mStride * 13, (size_t)R.mStride * 13 * sizeof(double));
          D.Add(&R.mEffectiveLength[(size_t)k * 3], 3 * sizeof(double));
          const TSampleDataInfo& Info = R.mInfo[(size_t)k];
          D.Add(&Info.mPeakValue, sizeof(float)); D.Add(&Info.mRmsValue, sizeof(float));
          D.Add(&Info.mDataOffset, sizeof(int)); D.Add(&Info.mNumberOfSamples, sizeof(int64_t));
          if (!R.mRhythmOffset.empty() && R.mpRhythmOnsets) {
            const int64_t t0 = R.mRhythmOffset[(size_t)k], t1 = R.mRhythmOffset[(size_t)k + 1];
            D.Add(R.mpRhythmOnsets + t0 * 2, (size_t)(t1 - t0) * 2 * sizeof(double));
            D.Add(R.mpRhythmScalars + (size_t)k * 14, 14 * sizeof(double));
            D.Add(R.mpRhythmStatistics + (size_t)k * 26, 26 * sizeof(double));
          }
          Total.mRowDigests[(size_t)(Done.mFiles[i] - Files.data())] = D.mHash ? D.mHash : 1;
        }
      }
      // with a database: the rows' column values are built here, by the eight workers, not by the one writer
      if (pPool && Options.mPrepareRowsInWorkers) {
        Done.mpRows = RowPool.Acquire();
        if (Done.mpRows->mRows.size() < Decoded.size()) Done.mpRows->mRows.resize(Decoded.size());
        for (size_t i = 0; i < n; ++i) {
          const int k = Done.mBatchIndex[i];
          if (k < 0 || !Done.mFailed[i].empty()) continue;
          const TSampleDescriptors Results = Done.mResults.Descriptors(k);
          RefillLowLevelColumns(Done.mpRows->mRows[(size_t)k], Results, &Done.mResults.mInfo[(size_t)k]);
        }
      }
    }
    const double tGpu1 = Now(), cGpu1 = ThreadCpuSeconds();
    {
      std::lock_guard<std::mutex> Lock(StatMutex);
      PhaseSeconds[1] += tGpu1 - tGpu0;
      PhaseCpuSeconds[1] += cGpu1 - cGpu0;
      for (int k = 0; k < 3; ++k) GpuSeconds[k] += Done.mResults.mSeconds[k];
      Total.mFiles += (int64_t)n;
      Total.mBatches += 1;
      Total.mFrames += Frames;
      Total.mPcmBytes += Work.mDecoded.empty() ? 0 : PcmBytes;
      Total.mResultBytes += ResultBytes;
      Total.mFilesPerDevice[(size_t)d] += (int64_t)n;
      Total.mPcmBytesPerDevice[(size_t)d] += Work.mDecoded.empty() ? 0 : PcmBytes;
    }
    Queue.Push(std::move(Work.mpDone));
    {
      std::lock_guard<std::mutex> Lock(StatMutex);
      Total.mSecondsPerDevice[(size_t)d] = Now() - Start;
    }
  };

  auto Worker = [&](int d) {
    try {
      struct TStagingLease {   // the worker's page-locked staging buffer goes back to the crawler when the worker ends
        TPinnedPool& mPool;
        std::unique_ptr<TPinned> mp;
        ~TStagingLease() { mPool.Release(std::move(mp)); }
      } Lease{StagingPool, StagingPool.Acquire(0)};
      TPinned& Staging = *Lease.mp;
      for (;;) {
        if (Abort) return;
        if (Options.mpAbortRequested && Options.mpAbortRequested->load()) { Stopped = true; return; }
        // the next FilesPerBatch files of the shard, or fewer when their bytes reach the batch's budget (long files:
        // the staging buffer, the device workspace and the result buffers all scale with the PCM of a batch)
        size_t Begin, End;
        int64_t BatchBytes = 0;
        TWork Work;
        {
          std::lock_guard<std::mutex> Lock(CursorMutex[(size_t)d]);
          const std::vector<const TCrawlFile*>& Mine = Shard[(size_t)d];
          Begin = End = Cursor[(size_t)d];
          if (Begin >= Mine.size()) return;
          while (End < Mine.size() && End - Begin < (size_t)FilesPerBatch) {
            const int64_t Size = FileBytes(*Mine[End]);
            if (End > Begin && BatchBytes + Size > BytesPerBatch) break;
            BatchBytes += Size;
            ++End;
          }
          Cursor[(size_t)d] = End;
          Work.mOrdinal = NextOrdinal.fetch_add(1);   // batches in the order they were cut (per device: the order of its files)
        }
        Work.mpDone.reset(new TFinishedBatch);
        TFinishedBatch& Done = *Work.mpDone;
        const size_t n = End - Begin;
        Done.mFiles.assign(Shard[(size_t)d].begin() + (long)Begin, Shard[(size_t)d].begin() + (long)End);
        Done.mProperties.resize(n);
        Done.mFailed.assign(n, std::string());
        Done.mSkipped.assign(n, 0);
        Done.mBatchIndex.assign(n, -1);
        // parse; lay the data chunks out in the page-locked staging buffer
        const double tParse0 = Now(), cParse0 = ThreadCpuSeconds();
        // Every file's samples go straight to their place in the page-locked staging buffer (the device arena's layout:
        // payloads back to back, 16-byte aligned, so that the C-ABI uploads the batch in one transfer): a memcpy out of
        // a file image, a pread out of the page cache for a file on disk.  The buffer is sized from the files' sizes
        // up front; only 8-bit files (widened to int16) can make it grow on the way.
        std::vector<TDecodedSample>& Decoded = Work.mDecoded;
        std::vector<size_t> Offset;
        size_t Bytes = 0;
        Staging.Reserve((size_t)BatchBytes + 16 * n + 64);
        TWaveFile Wave;
        for (size_t i = 0; i < n; ++i) {
          try {
            const TCrawlFile& f = *Done.mFiles[i];
            if (f.mpImage) Wave.OpenForRead(f.mpImage, f.mImageSize, f.mFileName);
            else Wave.OpenForRead(f.mFileName);
            if (!Options.mResample && Wave.SamplingRate() != Options.mSampleRate) { Done.mSkipped[i] = 1; Wave.Close(); continue; }
            TDecodedSample s = Wave.DescribeSample();
            const size_t Size = Wave.SampleDataBytes();
            if (Bytes + Size + 64 > Staging.mBytes) Staging.Grow(Bytes + Size + 64, Bytes);
            Wave.ReadSampleData((char*)Staging.mp + Bytes);
            Wave.Close();
            TFileProperties& p = Done.mProperties[i];
            p.mFileType = "wav";
            p.mFileSize = (int)Wave.FileSizeInBytes();
            p.mFileLength = (double)Wave.NumSamples() / (double)Wave.SamplingRate();
            p.mFileSampleRate = Wave.SamplingRate();
            p.mFileChannelCount = Wave.NumChannels();
            p.mFileBitDepth = Wave.BitsPerSample();
            Done.mBatchIndex[i] = (int)Decoded.size();
            Decoded.push_back(s);
            Offset.push_back(Bytes);
            Bytes += (Size + 15) & ~(size_t)15;
          } catch (const TReadableException& e) {
            Wave.Close();
            Done.mFailed[i] = std::string("Sample failed to load: ") + e.what();   // SampleAnalyser.cpp:372-387
          }
        }
        for (size_t k = 0; k < Decoded.size(); ++k) Decoded[k].mpInterleavedSamples = (char*)Staging.mp + Offset[k];
        {
          std::lock_guard<std::mutex> Lock(StatMutex);
          PhaseSeconds[0] += Now() - tParse0;
          PhaseCpuSeconds[0] += ThreadCpuSeconds() - cParse0;
        }
        // GPU: LoadSample + descriptors + statistics; results straight into page-locked buffers
        Process(d, Work, 0);
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> Lock(StatMutex);
      if (FirstError.empty()) FirstError = e.what();
      Abort = true;
    }
  };

  // the single writer (SampleAnalyser.cpp:413-415: one mutex around the pool)
  std::thread Writer([&] {
    while (std::unique_ptr<TFinishedBatch> p = Queue.Pop()) {
      const double t0 = Now(), c0 = ThreadCpuSeconds();
      int64_t Failed = 0, Skipped = 0;
      // Only whole batches reach the database.  A crawl that is ending (Abort: a lost device, an exception in a worker, an
      // earlier failed insert) writes nothing of the batches still queued; a batch that has begun is finished and
      // committed unless one of its OWN inserts fails -- that insert has rolled the batch's transaction back
      // (SqlitePool.cpp: InsertColumns / InsertFailedSample), and nothing more of the batch is written behind it.
      // (Until round 5 the loop also broke when another thread set Abort mid-batch, and the commit below then committed
      // the partial batch.)
      bool WriteBatch = !Abort, InsertFailed = false;
      try {
        if (pPool && WriteBatch) pPool->BeginTransaction();     // one commit per batch of files; the rows are those of one commit per file
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> Lock(StatMutex);
        if (FirstError.empty()) FirstError = e.what();
        Abort = true;
        WriteBatch = false;
      }
      for (size_t i = 0; i < p->mFiles.size() && WriteBatch && !InsertFailed; ++i) {
        const TCrawlFile& f = *p->mFiles[i];
        try {
          if (p->mSkipped[i]) {
            ++Skipped;
          } else if (!p->mFailed[i].empty()) {
            ++Failed;
            if (pPool) pPool->InsertFailedSample(f.mFileName, f.mModificationTime, p->mFailed[i]);
          } else if (pPool) {
            const int k = p->mBatchIndex[i];
            if (p->mpRows) {
              pPool->InsertColumns(f.mFileName, f.mModificationTime, p->mProperties[i], p->mpRows->mRows[(size_t)k]);
            } else {
              const TSampleDescriptors Results = p->mResults.Descriptors(k);
              pPool->InsertSample(f.mFileName, f.mModificationTime, p->mProperties[i], Results, &p->mResults.mInfo[(size_t)k]);
            }
          }
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> Lock(StatMutex);
          if (FirstError.empty()) FirstError = e.what();
          Abort = true;
          InsertFailed = true;
        }
      }
      try {
        if (pPool && WriteBatch && !InsertFailed) pPool->CommitTransaction();
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> Lock(StatMutex);
        if (FirstError.empty()) FirstError = e.what();
        Abort = true;
      }
      Pool.Release(std::move(p->mpRecords));
      StatisticsPool.Release(std::move(p->mpStatistics));
      RhythmPool.Release(std::move(p->mpRhythm));
      RowPool.Release(std::move(p->mpRows));
      std::lock_guard<std::mutex> Lock(StatMutex);
      Total.mFailedFiles += Failed;
      Total.mSkippedSampleRateFiles += Skipped;
      if (pPool) Total.mWriterSeconds += Now() - t0;
      PhaseCpuSeconds[2] += ThreadCpuSeconds() - c0;
    }
  });

  const double CpuStart = ProcessCpuSeconds();
  std::vector<std::thread> Workers;
  for (int d = 0; d < G; ++d)
    for (int w = 0; w < W; ++w) Workers.emplace_back(Worker, d);
  for (std::thread& t : Workers) t.join();
  Queue.Close();
  Writer.join();
  Total.mSeconds = Now() - Start;
  Total.mCpuSeconds = ProcessCpuSeconds() - CpuStart;
  Total.mAborted = Stopped.load();
  if (std::getenv("AFEC_CRAWL_TIMING"))
    std::fprintf(stderr, "[afec crawl] %.1f ms wall; worker time summed over %d workers: parse + staging %.1f ms, GPU round trip %.1f ms\n",
                 Total.mSeconds * 1e3, G * W, PhaseSeconds[0] * 1e3, PhaseSeconds[1] * 1e3);
  if (std::getenv("AFEC_CRAWL_TIMING"))
    std::fprintf(stderr, "[afec crawl]   round trip = create (upload, LoadSample) %.1f ms + enqueue %.1f ms + fetch (wait, download) %.1f ms\n",
                 GpuSeconds[0] * 1e3, GpuSeconds[1] * 1e3, GpuSeconds[2] * 1e3);
  if (std::getenv("AFEC_CRAWL_TIMING"))
    std::fprintf(stderr, "[afec crawl]   CPU %.1f ms (%.2f busy CPUs): workers parse + staging %.1f ms, workers GPU round trip %.1f ms, writer %.1f ms, other threads %.1f ms\n",
                 Total.mCpuSeconds * 1e3, Total.mCpuSeconds / Total.mSeconds, PhaseCpuSeconds[0] * 1e3, PhaseCpuSeconds[1] * 1e3,
                 PhaseCpuSeconds[2] * 1e3, (Total.mCpuSeconds - PhaseCpuSeconds[0] - PhaseCpuSeconds[1] - PhaseCpuSeconds[2]) * 1e3);
  if (!FirstError.empty()) throw TReadableException(FirstError);
  return Total;
}

}  // namespace afec

namespace {
std::mutex gCrawlerMutex;
std::atomic<int64_t> gBytesPerBatch(0);   // afec_crawl_set_bytes_per_batch: 0 = TCrawlOptions' default
std::vector<std::pair<std::string, afec::TCrawler*>> gCrawlers;   // never destroyed at exit: the HIP runtime may be gone by then
}  // namespace

extern "C" void afec_crawl_release(void) {
  std::lock_guard<std::mutex> Lock(gCrawlerMutex);
  for (auto& Entry : gCrawlers) delete Entry.second;
  gCrawlers.clear();
}

extern "C" void afec_crawl_set_bytes_per_batch(int64_t bytes) { gBytesPerBatch = bytes; }

namespace {
std::mutex gPragmaMutex;
std::string gDatabasePragmas;
std::atomic<bool> gResample(true);
std::atomic<int> gTestFailBatch(-1), gTestFailAttempts(0), gTestDeviceLost(0);
std::atomic<int64_t> gDeviceBytesPerBatch(0);
std::atomic<int> gFrameKernel(-1);   // afec_crawl_set_frame_kernel: -1 = TCrawlOptions' default
std::atomic<bool> gAbortRequested(false);   // afec_crawl_request_abort
}  // namespace
extern "C" void afec_crawl_request_abort(void) { gAbortRequested = true; }
extern "C" void afec_crawl_set_frame_kernel(int32_t frame_kernel) { gFrameKernel = frame_kernel; }
extern "C" void afec_crawl_set_test_fault(int32_t batch, int32_t attempts, int32_t device_lost) {
  gTestFailBatch = batch; gTestFailAttempts = attempts; gTestDeviceLost = device_lost;
}
extern "C" void afec_crawl_set_device_bytes_per_batch(int64_t bytes) { gDeviceBytesPerBatch = bytes; }
extern "C" void afec_crawl_set_resample(int32_t resample) { gResample = resample != 0; }
extern "C" void afec_crawl_set_database_pragmas(const char* pragmas) {
  std::lock_guard<std::mutex> Lock(gPragmaMutex);
  gDatabasePragmas = pragmas ? pragmas : "";
}

extern "C" int afec_crawl_wave_images(const char* const* names, const void* const* images, const int64_t* sizes, int32_t n_files,
                                      const int32_t* devices, int32_t n_devices, int32_t workers_per_device,
                                      int32_t files_per_batch, const char* database_path, double* stats, char* error,
                                      int32_t error_size) {
  return afec_crawl_wave_images_ex(names, images, sizes, n_files, devices, n_devices, workers_per_device, files_per_batch,
                                   database_path, stats, nullptr, nullptr, nullptr, error, error_size);
}

extern "C" int afec_crawl_wave_images_ex(const char* const* names, const void* const* images, const int64_t* sizes,
                                         int32_t n_files, const int32_t* devices, int32_t n_devices, int32_t workers_per_device,
                                         int32_t files_per_batch, const char* database_path, double* stats,
                                         double* device_stats, uint64_t* row_digests, double* crawl_facts, char* error,
                                         int32_t error_size) {
  try {
    std::vector<afec::TCrawlFile> Files((size_t)n_files);
    for (int32_t i = 0; i < n_files; ++i) {
      Files[(size_t)i].mFileName = names[i];
      Files[(size_t)i].mModificationTime = 1700000000 + i;
      Files[(size_t)i].mpImage = images ? images[i] : nullptr;
      Files[(size_t)i].mImageSize = images ? (size_t)sizes[i] : 0;
    }
    afec::TCrawlOptions Options;
    Options.mDevices.assign(devices, devices + n_devices);
    if (workers_per_device > 0) Options.mWorkersPerDevice = workers_per_device;
    if (files_per_batch > 0) Options.mFilesPerBatch = files_per_batch;
    if (database_path) Options.mDatabasePath = database_path;
    if (gBytesPerBatch > 0) Options.mBytesPerBatch = gBytesPerBatch;
    Options.mResample = gResample;
    Options.mTestFailBatch = gTestFailBatch; Options.mTestFailAttempts = gTestFailAttempts; Options.mTestDeviceLost = gTestDeviceLost != 0;
    if (gDeviceBytesPerBatch > 0) Options.mDeviceBytesPerBatch = gDeviceBytesPerBatch;
    if (gFrameKernel >= 0) Options.mFrameKernel = gFrameKernel;
    Options.mRowDigests = row_digests != nullptr;
    Options.mpAbortRequested = &gAbortRequested;
    {
      std::lock_guard<std::mutex> Lock(gPragmaMutex);
      Options.mDatabasePragmas = gDatabasePragmas;
    }
    // one crawler per (devices, geometry), kept between calls
    // (the registry lock is held for the whole crawl: afec_crawl_release cannot delete a crawler that is in use, and
    // crawls through this entry point run one at a time)
    std::lock_guard<std::mutex> Lock(gCrawlerMutex);
    gAbortRequested = false;
    afec::TCrawler* pCrawler = nullptr;
    {
      std::string Key;
      for (int d : Options.mDevices) Key += std::to_string(d) + ",";
      Key += "k" + std::to_string(Options.mFrameKernel);   // a crawler keeps the kernel layout its plans were built with
      for (auto& Entry : gCrawlers)
        if (Entry.first == Key) pCrawler = Entry.second;
      if (!pCrawler) {
        pCrawler = new afec::TCrawler(Options);
        gCrawlers.emplace_back(Key, pCrawler);
      }
    }
    const afec::TCrawlStatistics s = pCrawler->Crawl(Files, Options);
    if (stats) {
      stats[0] = (double)s.mFiles; stats[1] = (double)s.mFailedFiles; stats[2] = (double)s.mFrames; stats[3] = (double)s.mPcmBytes;
      stats[4] = (double)s.mResultBytes; stats[5] = s.mSeconds; stats[6] = s.mWriterSeconds;
      stats[7] = (double)s.mBatches;
      for (int32_t d = 0; d < n_devices; ++d) stats[8 + d] = (double)s.mFilesPerDevice[(size_t)d];
      stats[8 + n_devices] = s.mCpuSeconds;
      stats[9 + n_devices] = (double)s.mSkippedSampleRateFiles;
      stats[10 + n_devices] = (double)s.mRetriedBatches;
      stats[11 + n_devices] = (double)s.mDeviceFailedFiles;
    }
    if (device_stats)
      for (int32_t d = 0; d < n_devices; ++d) {
        device_stats[3 * d] = (double)s.mFilesPerDevice[(size_t)d];
        device_stats[3 * d + 1] = (double)s.mPcmBytesPerDevice[(size_t)d];
        device_stats[3 * d + 2] = s.mSecondsPerDevice[(size_t)d];
      }
    if (row_digests)
      for (int32_t i = 0; i < n_files; ++i) row_digests[i] = s.mRowDigests[(size_t)i];
    if (crawl_facts) { crawl_facts[0] = (double)s.mWorkersPerDevice; crawl_facts[1] = s.mUsableHostCpus; crawl_facts[2] = s.mAborted ? 1.0 : 0.0; }
    return 0;
  } catch (const std::exception& e) {
    if (error && error_size > 0) std::snprintf(error, (size_t)error_size, "%s", e.what());
    return -1;
  }
}

namespace {
int WaveProbe(afec::TWaveFile& Wave, int64_t* props, void* payload, int64_t payload_capacity) {
  const afec::TDecodedSample s = Wave.DescribeSample();
  const int64_t Bytes = (int64_t)Wave.SampleDataBytes();
  props[0] = Wave.NumChannels(); props[1] = Wave.SamplingRate(); props[2] = Wave.BitsPerSample();
  props[3] = (int64_t)Wave.SampleType(); props[4] = Wave.NumSamples(); props[5] = s.mFormat; props[6] = Bytes;
  if (payload && Bytes <= payload_capacity) Wave.ReadSampleData(payload);
  return 0;
}
}  // namespace

extern "C" int afec_wave_probe(const void* image, int64_t size, int64_t* props, void* payload, int64_t payload_capacity,
                               char* error, int32_t error_size) {
  try {
    afec::TWaveFile Wave;
    Wave.OpenForRead(image, (size_t)size);
    return WaveProbe(Wave, props, payload, payload_capacity);
  } catch (const std::exception& e) {
    if (error && error_size > 0) std::snprintf(error, (size_t)error_size, "%s", e.what());
    return -1;
  }
}

extern "C" int afec_wave_probe_file(const char* path, int64_t* props, void* payload, int64_t payload_capacity, char* error,
                                    int32_t error_size) {
  try {
    afec::TWaveFile Wave;
    Wave.OpenForRead(std::string(path));
    return WaveProbe(Wave, props, payload, payload_capacity);
  } catch (const std::exception& e) {
    if (error && error_size > 0) std::snprintf(error, (size_t)error_size, "%s", e.what());
    return -1;
  }
}

extern "C" double afec_usable_host_cpus(void) { return afec::UsableHostCpus(); }
extern "C" int32_t afec_workers_per_device_for(int32_t n_devices) { return afec::WorkersPerDeviceFor(n_devices); }

extern "C" int afec_shard_of_file(int64_t file_index, int32_t n_devices) { return afec::ShardOfFile(file_index, n_devices); }
