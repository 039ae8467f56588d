// afec_amd/host/Crawler.h -- streaming, sharded host driver above the C-ABI: the analogue of the
// reference crawler's worker pool (Source/Crawler/XCrawler/Source/Crawler.cpp:599, 706-728: one self-contained
// task per file on a thread pool that shares one const analyser) and of its single database writer
// (SampleAnalyser.cpp:413-415), for GPUs:
//
//   * file i of a crawl belongs to device i mod G (ShardOfFile): no exchange between devices, no collective;
//   * every device is fed by W worker threads; a worker owns one page-locked staging buffer, fills it with the data
//     chunks of its next batch of WAV files (TWaveFile), and runs LoadSample + every per-frame descriptor + the
//     statistics on the GPU (afx_batch_create_from_raw / run); the C-ABI gives every batch its own streams and
//     workspace and sends the large transfers of a plan through one upload and one download stream, so the upload of
//     one worker's batch overlaps the kernels of another's and the download of a third (the crawler asks the HIP
//     runtime for 16 hardware queues for that and lets the workers' waits sleep: TCrawlOptions::mHardwareQueues,
//     mSleepingWaits);
//   * results come back as the raw per-frame records and statistics (one transfer each, afx_batch_fetch_records)
//     and go through a bounded queue to ONE writer, which materialises TSampleDescriptors per file and inserts them
//     into the reference's `assets` table (TSqliteSampleDescriptorPool) -- or just counts them.
#pragma once

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "SampleAnalyser.h"

namespace afec {

struct TCrawlFile {
  std::string mFileName;            // the database key (the crawler stores paths relative to the crawl root)
  int mModificationTime = 0;
  const void* mpImage = nullptr;    // the file's bytes when it is already in memory; nullptr: read mFileName
  size_t mImageSize = 0;
};

struct TCrawlOptions {
  std::vector<int> mDevices = {0};  // HIP device ordinals
  // host threads (= batches in flight) per device.  0: picked from the CPUs the process may really use (the cgroup CPU
  // bandwidth quota, else the affinity mask) -- floor(CPUs / devices), at most 5, at least 1 (WorkersPerDeviceFor): the
  // host link is full from 5 on (284 k one-second files/s; 8: 281 k) and every worker costs host CPU -- 1.78 busy CPUs per
  // GPU with 5, 1.90-2.2 with 8 (round 5, tools/thread_cpu.py) -- so a node with 8 GPUs and a 16-CPU quota gets 2 per GPU,
  // one GPU alone 5.  Crawls that read the files themselves (pread out of the page cache is the cost there) gain from more.
  int mWorkersPerDevice = 0;
  int mFilesPerBatch = 512;         // measured best on one MI355X for 1 s stereo files (profiles/r02/README.md)
  int64_t mBytesPerBatch = 128 << 20;  // a batch also ends before the file that takes it over this many file bytes (a
                                    // 16-bit mono file needs ~10 x its size in device memory: PCM as doubles, spectra)
  std::string mDatabasePath;        // empty: results are counted, not stored
  std::string mDatabasePragmas;     // run when the database is opened ("" = sqlite's defaults, like the reference; SqlitePool.h)
  int mSampleRate = 44100, mFftFrameSize = 2048, mHopFrameSize = 1024;
  // Layout of the STFT kernel of the crawler's plans (afx_plan_desc.frame_kernel; TSampleAnalyser's FrameKernel).  Pinned:
  // the library's AUTO picks the layout by batch size, the two layouts round differently (their results agree to 1e-6,
  // the discrete descriptors may flip), and which batch a file lands in -- a crawl's tail batch, the halves of a retried
  // batch, a batch cut by mDeviceBytesPerBatch -- must not decide its database row.  1 (AFX_FRAME_KERNEL_WAVE64): one
  // frame per 64-lane wave, the better one up to ~30 000 frames per batch (512 one-second files are 20 000); 2
  // (AFX_FRAME_KERNEL_HALFWAVE) for crawls of long files in large batches.  A crawler keeps the layout it was built with.
  int mFrameKernel = 1;
  // Runtime knobs the crawler applies itself, to its own plans only (libafx_hip.so changes nothing process-wide):
  // * the HIP runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and the
  //   packets of a queue execute in order; with 2 streams per batch in flight + the plan's two copy streams the
  //   crawl wants 16 (205 k -> 268 k one-second files/s on one MI355X; more than the device's ~20 hardware slots are
  //   time-sliced: 175 k).  The runtime reads the variable when it initialises, so the first TCrawler of a process
  //   sets it -- unless the variable is already set or mHardwareQueues is 0 -- and it takes effect only when no HIP
  //   call was made before.
  // * worker threads that wait for the device sleep between event polls (20 us naps) instead of spinning: the same
  //   throughput with ~3 instead of ~7 busy CPUs per GPU.
  int mHardwareQueues = 16;
  bool mSleepingWaits = true;
  // files at another sampling rate than the analyser's are converted on the GPU like the reference converts them on
  // the CPU (SampleAnalyser.cpp:563-607, libresample; afx_resample.hip); false: they are skipped and counted
  bool mResample = true;
  // with a database: the workers build the rows' column values (msgpack BLOBs: ~12 % of the single writer's time per
  // row), the writer only binds and steps; false: the writer does both
  bool mPrepareRowsInWorkers = true;
  // a batch is also cut (before it is tried) when the device memory its files need -- their samples *after* the
  // sample-rate conversion, spectra, records -- exceeds this: a header that claims 1 kHz makes a small file a large one
  int64_t mDeviceBytesPerBatch = (int64_t)2 << 30;
  // Fault injection for the tests of the failure path (a batch whose GPU round trip fails is retried in halves, a single
  // file twice, then recorded as failed; the crawl goes on): the first mTestFailAttempts GPU attempts on files of the
  // mTestFailBatch-th batch of the crawl throw (-1: every attempt); mTestDeviceLost: and the device counts as lost
  // (the crawl must end with the error).
  int mTestFailBatch = -1, mTestFailAttempts = 0;
  bool mTestDeviceLost = false;
  // TCrawlStatistics::mRowDigests is filled: one 64-bit digest per file over everything the device returned for it
  // (per-frame records, statistics, rhythm results, LoadSample's facts), computed by the worker that analysed its batch.
  // For checks that the same content gives the same row whichever device / batch it landed on (bench.py's sharded crawl)
  bool mRowDigests = false;
  // Set from outside (another thread, a signal handler) to end the crawl early -- the reference's sAbortProcessing, which
  // its SIGINT handler sets and every task checks before it starts (Crawler.cpp:69-73, 717-720).  Workers take no new batch
  // once it reads true; batches already analysed are delivered and written whole, the crawl returns normally with
  // TCrawlStatistics::mAborted set and mFiles counting what was done.  nullptr: the crawl cannot be aborted from outside.
  const std::atomic<bool>* mpAbortRequested = nullptr;
};

// CPUs the process may use at once: the cgroup CPU bandwidth quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us) where
// one is set, else the size of the affinity mask
double UsableHostCpus();
// TCrawlOptions::mWorkersPerDevice == 0: floor(UsableHostCpus() / NumberOfDevices), at most 5, at least 1
int WorkersPerDeviceFor(int NumberOfDevices);

struct TCrawlStatistics {
  int64_t mFiles = 0, mFailedFiles = 0, mFrames = 0, mBatches = 0;
  // Files at another sampling rate than the analyser's that were left out because TCrawlOptions::mResample is off
  // (neither analysed nor written as failed samples: a later crawl finds them missing, not broken)
  int64_t mSkippedSampleRateFiles = 0;
  int64_t mPcmBytes = 0;            // bytes of PCM uploaded
  int64_t mResultBytes = 0;         // bytes of records + statistics downloaded
  double mSeconds = 0;              // first file read .. last result delivered to the writer
  double mWriterSeconds = 0;        // time the writer spent inserting (0 without a database)
  std::vector<int64_t> mFilesPerDevice;
  std::vector<int64_t> mPcmBytesPerDevice;   // bytes of PCM uploaded to each device
  std::vector<double> mSecondsPerDevice;     // crawl start .. the device's last batch handed to the writer
  int mWorkersPerDevice = 0;                 // the worker threads per device the crawl ran with
  double mUsableHostCpus = 0;                // UsableHostCpus() when the crawl started
  std::vector<uint64_t> mRowDigests;         // TCrawlOptions::mRowDigests: [file], 0 for files that were not analysed
  bool mAborted = false;                     // TCrawlOptions::mpAbortRequested became true before every file was taken
  double mCpuSeconds = 0;           // CPU time the process spent during the crawl (all threads): mCpuSeconds / mSeconds = busy CPUs
  // GPU_MAX_HW_QUEUES as the environment had it when the crawler was built (0: unset).  The HIP runtime reads it at its
  // first call: when the process had used HIP before, a value set here came too late (175-205 k instead of 268 k files/s)
  int mHardwareQueuesInEnvironment = 0;
  int64_t mRetriedBatches = 0;      // GPU round trips that failed on a live device and were retried (in halves / once more)
  int64_t mDeviceFailedFiles = 0;   // files recorded as "Sample failed to analyse: ..." after their own two attempts failed
};

// file i -> device index i mod G
inline int ShardOfFile(int64_t FileIndex, int NumberOfDevices) { return (int)(FileIndex % NumberOfDevices); }

// analyse every file; throws TReadableException when a device or the database cannot be set up or a device stops
// answering (single files that cannot be read or analysed -- also after a failed GPU round trip, retried in halves --
// are counted / inserted as failed samples, SampleAnalyser.cpp:368-408)
TCrawlStatistics CrawlWaveFiles(const std::vector<TCrawlFile>& Files, const TCrawlOptions& Options);

// The same with the set-up kept between crawls: the analysers (one plan per device and its pooled device workspaces)
// and the page-locked staging / result buffers.  Options of later crawls must name the same devices and geometry;
// workers, batch size and database may differ.  One crawl at a time per crawler.
class TCrawler {
public:
  explicit TCrawler(const TCrawlOptions& Options);
  ~TCrawler();
  TCrawler(const TCrawler&) = delete;
  TCrawler& operator=(const TCrawler&) = delete;
  TCrawlStatistics Crawl(const std::vector<TCrawlFile>& Files, const TCrawlOptions& Options);

private:
  struct TImpl;
  TImpl* mpImpl;
};

}  // namespace afec

extern "C" {
// C entry point of CrawlWaveFiles for callers without C++ (bench.py, tests): file images in memory, or -- images ==
// NULL -- files on disk, names[i] being their paths.  The process keeps
// one TCrawler per (devices, geometry) between calls (afec_crawl_release drops them), so a second crawl starts warm.
// stats: [files, failed, frames, pcm_bytes, result_bytes, seconds, writer_seconds, batches, files on device 0, 1, ...,
// cpu_seconds, files skipped for their sampling rate, retried batches, files failed on the device] (8 + n_devices + 4 doubles);
// returns 0, or -1 with the message in error.
int afec_crawl_wave_images(const char* const* names, const void* const* images, const int64_t* sizes, int32_t n_files,
                           const int32_t* devices, int32_t n_devices, int32_t workers_per_device, int32_t files_per_batch,
                           const char* database_path, double* stats, char* error, int32_t error_size);
// The same with the per-device figures and the row digests of a sharded crawl: device_stats (NULL or [n_devices][3]) =
// {files, bytes of PCM uploaded, seconds until the device's last batch was delivered} per device; row_digests (NULL or
// [n_files]) = TCrawlStatistics::mRowDigests; crawl_facts (NULL or [3]) = {worker threads per device the crawl ran with,
// UsableHostCpus(), 1 when afec_crawl_request_abort ended the crawl early}.  workers_per_device <= 0: TCrawlOptions' default (picked from the usable CPUs and the device count).
int afec_crawl_wave_images_ex(const char* const* names, const void* const* images, const int64_t* sizes, int32_t n_files,
                              const int32_t* devices, int32_t n_devices, int32_t workers_per_device, int32_t files_per_batch,
                              const char* database_path, double* stats, double* device_stats, uint64_t* row_digests,
                              double* crawl_facts, char* error, int32_t error_size);
// BASELINE.json configs[1]'s input: n floats U(-1, 1) from std::mt19937(seed) through
// std::uniform_real_distribution<float>(-1, 1) -- the generator the CPU-baseline driver built from the reference's own objects draws from
// (its `time` mode), so both sides of the comparison see the same stream
void afec_fill_uniform_mt19937(float* dst, int64_t n, uint32_t seed);
double afec_usable_host_cpus(void);                      // afec::UsableHostCpus
int32_t afec_workers_per_device_for(int32_t n_devices);  // afec::WorkersPerDeviceFor
// TWaveFile::OpenForRead on a file image: props = {channels, sampling rate, bits per sample, TSampleType, sample
// frames, AFX_RAW_* format of the decoded payload, payload bytes}; when payload is not NULL the decoded payload
// (8-bit files widened to int16) is copied there (payload_capacity bytes).  Returns 0, or -1 with the reader's message.
int afec_wave_probe(const void* image, int64_t size, int64_t* props /* [7] */, void* payload, int64_t payload_capacity,
                    char* error, int32_t error_size);
// the same for a file on disk (TWaveFile::OpenForRead(FileName): head parsed, data chunk read by pread)
int afec_wave_probe_file(const char* path, int64_t* props /* [7] */, void* payload, int64_t payload_capacity, char* error,
                         int32_t error_size);
int afec_shard_of_file(int64_t file_index, int32_t n_devices);
void afec_crawl_release(void);
// ends the crawl that is running through afec_crawl_wave_images / _ex early (TCrawlOptions::mpAbortRequested; the
// reference's SIGINT handler, Crawler.cpp:69-73): safe from another thread or a signal handler; cleared when a crawl starts
void afec_crawl_request_abort(void);
// TCrawlOptions::mBytesPerBatch of the crawls that follow (0: the default)
void afec_crawl_set_bytes_per_batch(int64_t bytes);
// TCrawlOptions::mDatabasePragmas of the crawls that follow (NULL or "": none)
void afec_crawl_set_database_pragmas(const char* pragmas);
// TCrawlOptions::mResample of the crawls that follow (0: files at another rate than the analyser's are skipped and counted)
void afec_crawl_set_resample(int32_t resample);
// TCrawlOptions::mTestFailBatch / mTestFailAttempts / mTestDeviceLost of the crawls that follow (-1, 0, 0: no fault)
void afec_crawl_set_test_fault(int32_t batch, int32_t attempts, int32_t device_lost);
// TCrawlOptions::mFrameKernel of the crawls that follow (-1: the default; 0 / 1 / 2 = AFX_FRAME_KERNEL_AUTO / WAVE64 / HALFWAVE)
void afec_crawl_set_frame_kernel(int32_t frame_kernel);
// TCrawlOptions::mDeviceBytesPerBatch of the crawls that follow (0: the default)
void afec_crawl_set_device_bytes_per_batch(int64_t bytes);
}
