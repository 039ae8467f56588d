// afec_amd/host/SqlitePool.h -- the reference's low-level descriptor database:
// TSqliteSampleDescriptorPool (Source/Crawler/FeatureExtraction/Export/SqliteSampleDescriptorPool.h,
// Source/SqliteSampleDescriptorPool.cpp:1290-1358 tables, 1582-1690 inserts) for the descriptors this
// library produces.  `PRAGMA user_version = 2`, one `assets` table with the reference's columns; columns of
// descriptors that are not computed here (rhythm_*) stay NULL.
//
// sqlite is bound at run time (dlopen of libsqlite3.so.0: the image ships the library but not its headers);
// the constructor throws TReadableException when it is not there.
#pragma once

#include <string>

#include "DescriptorColumns.h"
#include "SampleAnalyser.h"

namespace afec {

// the "Basic Data" of TSampleDescriptors the analyser cannot know (SampleAnalyser.cpp:730-746)
struct TFileProperties {
  std::string mFileType;   // lower-case extension
  int mFileSize = 0;       // bytes
  double mFileLength = 0;  // seconds
  int mFileSampleRate = 0, mFileChannelCount = 0, mFileBitDepth = 0;
};

class TSqliteSampleDescriptorPool {
public:
  // Pragmas: statements run right after the database is opened, before any table exists ("" = none: sqlite's defaults,
  // like the reference).  They tune how sqlite writes, not what: e.g. "PRAGMA page_size=65536; PRAGMA
  // journal_mode=MEMORY; PRAGMA synchronous=OFF" (rows of ~68 KB otherwise span 17 overflow pages of 4 KB; the
  // rollback journal and the fsync per commit go away) -- the file stays a plain sqlite database with the same table.
  explicit TSqliteSampleDescriptorPool(const std::string& DatabasePath, const std::string& Pragmas = std::string());
  ~TSqliteSampleDescriptorPool();
  TSqliteSampleDescriptorPool(const TSqliteSampleDescriptorPool&) = delete;
  TSqliteSampleDescriptorPool& operator=(const TSqliteSampleDescriptorPool&) = delete;

  // A caller that inserts many files in a row may put them into one transaction (the crawler's writer: one per batch
  // of files): the rows are identical, the commits (journal + fsync) are not paid per file.  Without these calls every
  // insert is its own transaction like the reference's (SqliteSampleDescriptorPool.cpp:1591-1640, 1655-1690).  A
  // failing insert rolls the open transaction back and throws.
  void BeginTransaction();
  void CommitTransaction();

  // INSERT OR REPLACE of one analysed file (in its own transaction unless one is open), status "succeeded"
  void InsertSample(const std::string& FileName, int ModificationTime, const TFileProperties& File,
                    const TSampleDescriptors& Results, const TSampleDataInfo* pInfo = nullptr);
  // The same in two steps, for pipelines whose worker threads build the column values (msgpack BLOBs, statistics) so
  // that the one writer thread only binds and steps: Values = RefillLowLevelColumns / LowLevelColumns of the file's
  // descriptors (DescriptorColumns.h; any thread), then InsertColumns on the writer.  The row is InsertSample's.
  void InsertColumns(const std::string& FileName, int ModificationTime, const TFileProperties& File,
                     const std::vector<TColumn>& Values);
  // the row of a file that could not be analysed: status "error: <Reason>", every descriptor NULL
  void InsertFailedSample(const std::string& FileName, int ModificationTime, const std::string& Reason);

private:
  struct TImpl;
  TImpl* mpImpl;
};

}  // namespace afec
