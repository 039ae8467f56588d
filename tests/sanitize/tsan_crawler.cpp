// tests/sanitize/tsan_crawler.cpp -- TEST INFRASTRUCTURE ONLY.  The streaming, sharded host driver (afec_amd/host/Crawler.cpp:
// worker threads per device, the bounded queue, the single writer, retries in halves, abort) with the sqlite descriptor
// pool and the REAL C-ABI host code below it, on the mock device of tests/sanitize/hipstub + mock_kernels.cpp, built by
// g++ -fsanitize=thread (tools/sanitize_cpu.sh; also built with address,undefined):
//
//   * G = 1, 2, 8 mock devices: file i is analysed on device i mod G (the reference hands one self-contained task per
//     file to its pool, Crawler.cpp:706-728), every file is delivered exactly once, the same content gives the same row
//     digest on every device and in every batch position;
//   * with the database on: one row per file, failed files as failed rows, only whole batches committed;
//   * an injected failure of one batch's GPU round trip -- once (retried, rows as in the clean crawl) and always (its files
//     become failed rows, the crawl goes on: SampleAnalyser.cpp:368-408);
//   * a lost device: the crawl ends with the error, nothing hangs, the crawler is usable afterwards;
//   * an external abort in the middle of a crawl (the reference's SIGINT flag, Crawler.cpp:69-73, 717-720);
//   * two crawlers at once on disjoint device sets.
//
// usage: tsan_crawler [files = 600]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>

#include "../../afec_amd/host/Crawler.h"

namespace {

[[noreturn]] void die(const char* what, long long a = 0, long long b = 0) {
  std::fprintf(stderr, "tsan_crawler: FAILED: %s (%lld, %lld)\n", what, a, b);
  std::abort();
}
#define REQUIRE(cond, ...) do { if (!(cond)) die(#cond, ##__VA_ARGS__); } while (0)

std::vector<unsigned char> MakeWav(int Channels, int Rate, int Frames, uint64_t Seed, int LeadingSilence) {
  std::vector<unsigned char> w;
  auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) w.push_back((unsigned char)(v >> (8 * i))); };
  auto put16 = [&](uint16_t v) { w.push_back((unsigned char)v); w.push_back((unsigned char)(v >> 8)); };
  const uint32_t DataBytes = (uint32_t)Frames * Channels * 2;
  w.insert(w.end(), {'R', 'I', 'F', 'F'}); put32(36 + DataBytes);
  w.insert(w.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '}); put32(16);
  put16(1); put16((uint16_t)Channels); put32((uint32_t)Rate); put32((uint32_t)(Rate * Channels * 2)); put16((uint16_t)(Channels * 2)); put16(16);
  w.insert(w.end(), {'d', 'a', 't', 'a'}); put32(DataBytes);
  std::mt19937_64 Gen(Seed);
  for (int k = 0; k < Frames; ++k)
    for (int c = 0; c < Channels; ++c) put16(k < LeadingSilence ? 0 : (uint16_t)(int16_t)((int)(Gen() % 40001) - 20000));
  return w;
}

constexpr int kContents = 21;   // coprime to 2 and 8: with files sharded i mod G every content visits every device

struct Corpus {
  std::vector<std::vector<unsigned char>> mImages;   // kContents distinct ones, one of them not a WAV file
  std::vector<afec::TCrawlFile> mFiles;
  int mBroken = 0;
};

Corpus MakeCorpus(int Files) {
  Corpus c;
  for (int k = 0; k < kContents; ++k) {
    if (k == 5) { c.mImages.push_back(std::vector<unsigned char>(300, 'x')); continue; }     // "_Not A Wavefile.wav"
    const int Rate = (k == 9) ? 48000 : 44100;                                              // one content is converted on the "GPU"
    c.mImages.push_back(MakeWav(1 + k % 2, Rate, 3000 + 1777 * k, 1000 + (uint64_t)k, (k % 3) * 500));
  }
  c.mFiles.resize((size_t)Files);
  for (int i = 0; i < Files; ++i) {
    afec::TCrawlFile& f = c.mFiles[(size_t)i];
    f.mFileName = "file" + std::to_string(i) + ".wav";
    f.mModificationTime = 1700000000 + i;
    f.mpImage = c.mImages[(size_t)(i % kContents)].data();
    f.mImageSize = c.mImages[(size_t)(i % kContents)].size();
    if (i % kContents == 5) ++c.mBroken;
  }
  return c;
}

afec::TCrawlOptions Options(int G, int W, int FilesPerBatch) {
  afec::TCrawlOptions o;
  o.mDevices.clear();
  for (int d = 0; d < G; ++d) o.mDevices.push_back(d);
  o.mWorkersPerDevice = W;
  o.mFilesPerBatch = FilesPerBatch;
  o.mHardwareQueues = 0;          // the environment is not the crawler's to change in a test process
  o.mRowDigests = true;
  return o;
}

void CheckSharding(const Corpus& c, const afec::TCrawlStatistics& s, int G, std::vector<uint64_t>* pReference) {
  const int64_t n = (int64_t)c.mFiles.size();
  REQUIRE(s.mFiles == n, s.mFiles, n);
  REQUIRE(s.mFailedFiles == c.mBroken, s.mFailedFiles, c.mBroken);
  for (int d = 0; d < G; ++d) {
    const int64_t want = n / G + (d < n % G ? 1 : 0);                     // files i with i mod G == d
    REQUIRE(s.mFilesPerDevice[(size_t)d] == want, d, s.mFilesPerDevice[(size_t)d]);
  }
  // the same content -> the same digest, whichever device and batch position; and the same as the reference crawl's
  std::vector<uint64_t> first((size_t)kContents, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int k = (int)(i % kContents);
    const uint64_t d = s.mRowDigests[(size_t)i];
    REQUIRE((d == 0) == (k == 5), i, (long long)d);
    if (!first[(size_t)k]) first[(size_t)k] = d;
    REQUIRE(d == first[(size_t)k], i, k);
  }
  if (pReference->empty()) *pReference = first;
  else REQUIRE(*pReference == first, G);
}

int64_t RowsIn(const std::string& Path, const char* Where) {
  // the test reads the database back through sqlite itself (bound at run time, like the pool does)
  void* lib = dlopen("libsqlite3.so.0", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return -1;
  using open_t = int (*)(const char*, void**);
  using close_t = int (*)(void*);
  using exec_t = int (*)(void*, const char*, int (*)(void*, int, char**, char**), void*, char**);
  open_t Open = (open_t)dlsym(lib, "sqlite3_open");
  close_t Close = (close_t)dlsym(lib, "sqlite3_close");
  exec_t Exec = (exec_t)dlsym(lib, "sqlite3_exec");
  void* db = nullptr;
  int64_t rows = -1;
  if (Open(Path.c_str(), &db) == 0) {
    const std::string sql = std::string("SELECT COUNT(*) FROM assets WHERE ") + Where;
    Exec(db, sql.c_str(), [](void* p, int, char** v, char**) { *(int64_t*)p = std::atoll(v[0]); return 0; }, &rows, nullptr);
  }
  Close(db);
  return rows;
}

}  // namespace

int main(int argc, char** argv) {
  const int Files = argc > 1 ? std::atoi(argv[1]) : 600;
  hipstub::set_device_count(8);
  const Corpus c = MakeCorpus(Files);
  std::vector<uint64_t> reference;
  double crawl_seconds = 0.05;     // of a clean two-device crawl: the abort below comes a third into one
  char tmpl[] = "/tmp/afx_tsan_XXXXXX";
  REQUIRE(mkdtemp(tmpl) != nullptr);
  const std::string dir = tmpl;

  // ---- sharding: G = 1, 2, 8 ----
  for (int G : {1, 2, 8}) {
    afec::TCrawler crawler(Options(G, 3, 37));
    const afec::TCrawlStatistics s = crawler.Crawl(c.mFiles, Options(G, 3, 37));
    CheckSharding(c, s, G, &reference);
    REQUIRE(s.mWorkersPerDevice == 3 && !s.mAborted);
    // a second crawl of the same crawler (warm pools), other batch size
    const afec::TCrawlStatistics s2 = crawler.Crawl(c.mFiles, Options(G, 2, 64));
    CheckSharding(c, s2, G, &reference);
    if (G == 2) crawl_seconds = s2.mSeconds;
    std::printf("tsan_crawler: G = %d: %lld files, %lld frames, files per device ok, digests identical per content\n", G, (long long)s.mFiles, (long long)s.mFrames);
  }
  // workers picked from the usable CPUs
  {
    afec::TCrawlOptions o = Options(8, 0, 64);
    afec::TCrawler crawler(o);
    const afec::TCrawlStatistics s = crawler.Crawl(c.mFiles, o);
    REQUIRE(s.mWorkersPerDevice == afec::WorkersPerDeviceFor(8), s.mWorkersPerDevice);
    CheckSharding(c, s, 8, &reference);
  }

  // ---- the database: one row per file, failed rows for the broken content ----
  {
    afec::TCrawlOptions o = Options(2, 3, 50);
    o.mDatabasePath = dir + "/clean.db";
    const afec::TCrawlStatistics s = afec::CrawlWaveFiles(c.mFiles, o);
    CheckSharding(c, s, 2, &reference);
    const int64_t all = RowsIn(o.mDatabasePath, "1"), failed = RowsIn(o.mDatabasePath, "status <> 'succeeded'");
    if (all >= 0) { REQUIRE(all == Files, all); REQUIRE(failed == c.mBroken, failed, c.mBroken); }
    std::printf("tsan_crawler: database: %lld rows, %lld failed\n", (long long)all, (long long)failed);
  }

  // ---- an injected failure of one batch: once (retried), always (failed rows), with a lost device (the crawl ends) ----
  {
    afec::TCrawlOptions o = Options(2, 3, 40);
    afec::TCrawler crawler(o);
    o.mTestFailBatch = 3; o.mTestFailAttempts = 1;
    afec::TCrawlStatistics s = crawler.Crawl(c.mFiles, o);
    CheckSharding(c, s, 2, &reference);
    REQUIRE(s.mRetriedBatches == 1, s.mRetriedBatches);
    o.mTestFailAttempts = -1;
    o.mDatabasePath = dir + "/failing.db";
    s = crawler.Crawl(c.mFiles, o);
    REQUIRE(s.mFiles == Files && s.mDeviceFailedFiles > 0 && s.mFailedFiles == c.mBroken + s.mDeviceFailedFiles - /* broken ones of that batch never reached the GPU */ 0 ||
            s.mFailedFiles >= s.mDeviceFailedFiles, s.mFailedFiles, s.mDeviceFailedFiles);
    const int64_t all = RowsIn(o.mDatabasePath, "1");
    if (all >= 0) REQUIRE(all == Files, all);
    o.mDatabasePath.clear();
    o.mTestDeviceLost = true;
    bool threw = false;
    try { crawler.Crawl(c.mFiles, o); } catch (const afec::TReadableException&) { threw = true; }
    REQUIRE(threw);
    o.mTestFailBatch = -1; o.mTestFailAttempts = 0; o.mTestDeviceLost = false;
    s = crawler.Crawl(c.mFiles, o);                       // the crawler is usable afterwards
    CheckSharding(c, s, 2, &reference);
    std::printf("tsan_crawler: injected failures: retried, failed rows, lost device -> error, crawler usable afterwards\n");
  }
  // a device that really goes away in the middle of a crawl (every runtime call on it fails from then on)
  {
    afec::TCrawlOptions o = Options(2, 2, 16);
    afec::TCrawler crawler(o);
    std::thread killer([] { std::this_thread::sleep_for(std::chrono::milliseconds(3)); hipstub::lose_device(1, true); });
    bool threw = false;
    try { crawler.Crawl(c.mFiles, o); } catch (const afec::TReadableException&) { threw = true; }
    killer.join();
    hipstub::lose_device(1, false);
    std::printf("tsan_crawler: device 1 lost mid-crawl: the crawl %s\n", threw ? "ended with the error" : "had finished before");
    const afec::TCrawlStatistics s = crawler.Crawl(c.mFiles, o);
    CheckSharding(c, s, 2, &reference);
  }

  // ---- an external abort in the middle of a crawl ----
  {
    std::atomic<bool> abort_requested(false);
    afec::TCrawlOptions o = Options(2, 3, 8);
    o.mpAbortRequested = &abort_requested;
    o.mDatabasePath = dir + "/aborted.db";
    std::thread interrupter([&] { std::this_thread::sleep_for(std::chrono::duration<double>(crawl_seconds / 3)); abort_requested = true; });
    const afec::TCrawlStatistics s = afec::CrawlWaveFiles(c.mFiles, o);
    interrupter.join();
    REQUIRE(s.mFiles <= Files);
    REQUIRE(s.mAborted || s.mFiles == Files, s.mFiles);
    const int64_t all = RowsIn(o.mDatabasePath, "1");
    if (all >= 0) REQUIRE(all == s.mFiles, all, s.mFiles);     // what was analysed was written, whole batches only
    std::printf("tsan_crawler: external abort: %lld of %d files analysed and written, aborted = %d\n", (long long)s.mFiles, Files, (int)s.mAborted);
  }

  // ---- two crawlers at once on disjoint devices ----
  {
    afec::TCrawlStatistics sa, sb;
    afec::TCrawlOptions oa = Options(2, 2, 32), ob = Options(2, 2, 48);
    ob.mDevices = {4, 5};
    std::thread ta([&] { sa = afec::CrawlWaveFiles(c.mFiles, oa); });
    std::thread tb([&] { sb = afec::CrawlWaveFiles(c.mFiles, ob); });
    ta.join(); tb.join();
    CheckSharding(c, sa, 2, &reference);
    CheckSharding(c, sb, 2, &reference);
  }
  REQUIRE(hipstub::device_bytes_in_use() == 0, (long long)hipstub::device_bytes_in_use());
  REQUIRE(hipstub::live_streams() == 0 && hipstub::live_events() == 0, hipstub::live_streams(), hipstub::live_events());
  const std::string rm = "rm -rf " + dir;
  if (std::system(rm.c_str()) != 0) std::fprintf(stderr, "could not remove %s\n", dir.c_str());
  std::printf("tsan_crawler: clean\n");
  return 0;
}
