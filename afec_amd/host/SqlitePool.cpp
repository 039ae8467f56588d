// afec_amd/host/SqlitePool.cpp -- see SqlitePool.h.
#include "SqlitePool.h"

#include <dlfcn.h>
#include <unistd.h>

#include <cstdlib>
#include <map>

namespace afec {

namespace {

// the handful of sqlite3 entry points used, resolved from the system library at run time (public, stable C API)
struct TSqliteApi {
  void* mpLibrary = nullptr;
  int (*open)(const char*, void**) = nullptr;
  int (*close)(void*) = nullptr;
  int (*exec)(void*, const char*, int (*)(void*, int, char**, char**), void*, char**) = nullptr;
  int (*prepare_v2)(void*, const char*, int, void**, const char**) = nullptr;
  int (*bind_text)(void*, int, const char*, int, void (*)(void*)) = nullptr;
  int (*bind_int)(void*, int, int) = nullptr;
  int (*bind_double)(void*, int, double) = nullptr;
  int (*bind_blob)(void*, int, const void*, int, void (*)(void*)) = nullptr;
  int (*bind_null)(void*, int) = nullptr;
  int (*step)(void*) = nullptr;
  int (*finalize)(void*) = nullptr;
  const char* (*errmsg)(void*) = nullptr;
};
constexpr int kSqliteOk = 0, kSqliteDone = 101;
void (*const kSqliteTransient)(void*) = reinterpret_cast<void (*)(void*)>(-1);

template <typename F>
void Resolve(void* pLibrary, const char* pName, F& Function) {
  Function = reinterpret_cast<F>(dlsym(pLibrary, pName));
  if (!Function) throw TReadableException(std::string("sqlite3 symbol not found: ") + pName);
}

}  // namespace

struct TSqliteSampleDescriptorPool::TImpl {
  TSqliteApi mApi;
  void* mpDatabase = nullptr;
  std::vector<TColumnSpec> mSchema;

  void Check(int Result, const char* pWhat) {
    if (Result != kSqliteOk && Result != kSqliteDone)
      throw TReadableException(std::string(pWhat) + ": " + (mpDatabase ? mApi.errmsg(mpDatabase) : "sqlite error"));
  }
  void Execute(const std::string& Sql) { Check(mApi.exec(mpDatabase, Sql.c_str(), nullptr, nullptr, nullptr), Sql.c_str()); }
  // first column of the first row as an int (TDatabase::ExecuteScalarInt)
  int ExecuteScalarInt(const std::string& Sql) {
    int Value = 0;
    auto Callback = [](void* pUser, int Count, char** ppValues, char**) -> int {
      if (Count > 0 && ppValues[0]) *static_cast<int*>(pUser) = std::atoi(ppValues[0]);
      return 0;
    };
    Check(mApi.exec(mpDatabase, Sql.c_str(), Callback, &Value, nullptr), Sql.c_str());
    return Value;
  }
};

TSqliteSampleDescriptorPool::TSqliteSampleDescriptorPool(const std::string& DatabasePath) : mpImpl(new TImpl) {
  TSqliteApi& A = mpImpl->mApi;
  A.mpLibrary = dlopen("libsqlite3.so.0", RTLD_NOW | RTLD_LOCAL);
  if (!A.mpLibrary) {
    delete mpImpl;
    throw TReadableException("libsqlite3.so.0 is not available: the descriptor database cannot be written");
  }
  try {
    Resolve(A.mpLibrary, "sqlite3_open", A.open);
    Resolve(A.mpLibrary, "sqlite3_close", A.close);
    Resolve(A.mpLibrary, "sqlite3_exec", A.exec);
    Resolve(A.mpLibrary, "sqlite3_prepare_v2", A.prepare_v2);
    Resolve(A.mpLibrary, "sqlite3_bind_text", A.bind_text);
    Resolve(A.mpLibrary, "sqlite3_bind_int", A.bind_int);
    Resolve(A.mpLibrary, "sqlite3_bind_double", A.bind_double);
    Resolve(A.mpLibrary, "sqlite3_bind_blob", A.bind_blob);
    Resolve(A.mpLibrary, "sqlite3_bind_null", A.bind_null);
    Resolve(A.mpLibrary, "sqlite3_step", A.step);
    Resolve(A.mpLibrary, "sqlite3_finalize", A.finalize);
    Resolve(A.mpLibrary, "sqlite3_errmsg", A.errmsg);
    mpImpl->Check(A.open(DatabasePath.c_str(), &mpImpl->mpDatabase), "sqlite3_open");
    mpImpl->mSchema = LowLevelSchema();
    // TSqliteSampleDescriptorPool::InitializeDatabase (SqliteSampleDescriptorPool.cpp:1224-1358): an existing assets
    // table is kept only at the current version; a newer database is refused, an older one is thrown away
    constexpr int kCurrentVersion = 2;                   // Export/SqliteSampleDescriptorPool.h:58
    bool CreateNewTables = false;
    if (mpImpl->ExecuteScalarInt("SELECT count(name) FROM sqlite_master WHERE type='table' AND name='assets'") != 1) {
      CreateNewTables = true;
    } else {
      const int DatabaseVersion = mpImpl->ExecuteScalarInt("PRAGMA user_version");
      if (DatabaseVersion > kCurrentVersion)
        throw TReadableException("Unknown database version: " + std::to_string(DatabaseVersion) +
                                 ". The database maybe got created by a newer version of the crawler.");
      if (DatabaseVersion < kCurrentVersion) {
        CreateNewTables = true;
        // try trashing the entire file first, drop the table as the fallback
        A.close(mpImpl->mpDatabase);
        mpImpl->mpDatabase = nullptr;
        const bool DeleteSucceeded = (::unlink(DatabasePath.c_str()) == 0);
        mpImpl->Check(A.open(DatabasePath.c_str(), &mpImpl->mpDatabase), "sqlite3_open (upgrade)");
        if (!DeleteSucceeded) {
          mpImpl->Execute("DROP table 'assets'");
          mpImpl->Execute("VACUUM");
        }
      }
    }
    if (CreateNewTables) {
      // SqliteSampleDescriptorPool.cpp:1304-1352: version + assets table, in one transaction
      std::string Ddl = "CREATE TABLE IF NOT EXISTS assets(filename TEXT PRIMARY KEY,modtime INTEGER,status TEXT";
      for (const TColumnSpec& c : mpImpl->mSchema) Ddl += "," + c.mName + " " + c.mpSqliteType;
      Ddl += ")";
      mpImpl->Execute("BEGIN");
      mpImpl->Execute("PRAGMA user_version = '" + std::to_string(kCurrentVersion) + "'");
      mpImpl->Execute(Ddl);
      mpImpl->Execute("COMMIT");
    }
  } catch (...) {
    if (mpImpl->mpDatabase) A.close(mpImpl->mpDatabase);
    dlclose(A.mpLibrary);
    delete mpImpl;
    throw;
  }
}

TSqliteSampleDescriptorPool::~TSqliteSampleDescriptorPool() {
  mpImpl->mApi.close(mpImpl->mpDatabase);
  dlclose(mpImpl->mApi.mpLibrary);
  delete mpImpl;
}

void TSqliteSampleDescriptorPool::InsertSample(const std::string& FileName, int ModificationTime,
                                               const TFileProperties& File, const TSampleDescriptors& Results,
                                               const TSampleDataInfo* pInfo) {
  TImpl& I = *mpImpl;
  const std::vector<TColumn> Values = LowLevelColumns(Results, pInfo);
  std::map<std::string, const TColumn*> ByName;
  for (const TColumn& c : Values) ByName[c.mName] = &c;

  // SqliteSampleDescriptorPool.cpp:1591-1640: all keys, INSERT OR REPLACE, one transaction per file
  std::string Sql = "INSERT OR REPLACE into assets(filename,modtime,status";
  for (const TColumnSpec& c : I.mSchema) Sql += "," + c.mName;
  Sql += ") values(?,?,?";
  for (size_t i = 0; i < I.mSchema.size(); ++i) Sql += ",?";
  Sql += ")";
  I.Execute("BEGIN");
  void* pStatement = nullptr;
  try {
    I.Check(I.mApi.prepare_v2(I.mpDatabase, Sql.c_str(), -1, &pStatement, nullptr), "prepare");
    I.Check(I.mApi.bind_text(pStatement, 1, FileName.c_str(), -1, kSqliteTransient), "bind filename");
    I.Check(I.mApi.bind_int(pStatement, 2, ModificationTime), "bind modtime");
    I.Check(I.mApi.bind_text(pStatement, 3, "succeeded", -1, kSqliteTransient), "bind status");
    int Index = 4;
    for (const TColumnSpec& c : I.mSchema) {
      int r = kSqliteOk;
      const auto Found = ByName.find(c.mName);
      if (Found != ByName.end()) {
        const TColumn& v = *Found->second;
        r = (v.mType == TColumn::kReal) ? I.mApi.bind_double(pStatement, Index, v.mReal)
                                        : I.mApi.bind_blob(pStatement, Index, v.mBlob.data(), (int)v.mBlob.size(), kSqliteTransient);
      } else if (c.mName == "file_type_S") r = I.mApi.bind_text(pStatement, Index, File.mFileType.c_str(), -1, kSqliteTransient);
      else if (c.mName == "file_size_R") r = I.mApi.bind_int(pStatement, Index, File.mFileSize);
      else if (c.mName == "file_length_R") r = I.mApi.bind_double(pStatement, Index, File.mFileLength);
      else if (c.mName == "file_sample_rate_R") r = I.mApi.bind_int(pStatement, Index, File.mFileSampleRate);
      else if (c.mName == "file_channel_count_R") r = I.mApi.bind_int(pStatement, Index, File.mFileChannelCount);
      else if (c.mName == "file_bit_depth_R") r = I.mApi.bind_int(pStatement, Index, File.mFileBitDepth);
      else {
        // a descriptor this library does not compute: a well-formed placeholder, because the reference's reader
        // unpacks every BLOB column of a "succeeded" row (SqliteSampleDescriptorPool.cpp:1004-1014): an empty
        // msgpack array for the vectors, 0 for the scalars
        static const unsigned char kEmptyMsgpackArray[1] = {0x90};
        const std::string Type = c.mpSqliteType;
        if (Type == "BLOB") r = I.mApi.bind_blob(pStatement, Index, kEmptyMsgpackArray, 1, kSqliteTransient);
        else if (Type == "REAL") r = I.mApi.bind_double(pStatement, Index, 0.0);
        else r = I.mApi.bind_text(pStatement, Index, "", -1, kSqliteTransient);
      }
      I.Check(r, c.mName.c_str());
      ++Index;
    }
    I.Check(I.mApi.step(pStatement), "insert");
    I.mApi.finalize(pStatement);
    pStatement = nullptr;
    I.Execute("COMMIT");
  } catch (...) {
    if (pStatement) I.mApi.finalize(pStatement);
    I.mApi.exec(I.mpDatabase, "ROLLBACK", nullptr, nullptr, nullptr);
    throw;
  }
}

void TSqliteSampleDescriptorPool::InsertFailedSample(const std::string& FileName, int ModificationTime,
                                                     const std::string& Reason) {
  TImpl& I = *mpImpl;
  // SqliteSampleDescriptorPool.cpp:1655-1690
  I.Execute("BEGIN");
  void* pStatement = nullptr;
  try {
    I.Check(I.mApi.prepare_v2(I.mpDatabase, "INSERT OR REPLACE into assets(filename, modtime, status) values (?,?,?)", -1,
                              &pStatement, nullptr), "prepare");
    const std::string Status = "error: " + Reason;
    I.Check(I.mApi.bind_text(pStatement, 1, FileName.c_str(), -1, kSqliteTransient), "bind filename");
    I.Check(I.mApi.bind_int(pStatement, 2, ModificationTime), "bind modtime");
    I.Check(I.mApi.bind_text(pStatement, 3, Status.c_str(), -1, kSqliteTransient), "bind status");
    I.Check(I.mApi.step(pStatement), "insert");
    I.mApi.finalize(pStatement);
    pStatement = nullptr;
    I.Execute("COMMIT");
  } catch (...) {
    if (pStatement) I.mApi.finalize(pStatement);
    I.mApi.exec(I.mpDatabase, "ROLLBACK", nullptr, nullptr, nullptr);
    throw;
  }
}

}  // namespace afec
