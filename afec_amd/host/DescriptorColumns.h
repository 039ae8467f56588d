// afec_amd/host/DescriptorColumns.h -- the on-disk representation of the low-level descriptors this library
// produces (the data-format half): column names and value encodings exactly as the reference's
// TSqliteSampleDescriptorPool writes them into its `assets` table
// (Source/Crawler/FeatureExtraction/Source/SqliteSampleDescriptorPool.cpp:1313-1358 DDL, 1582-1651 insert):
//
//   <descriptor>_<R|VR|VVR>             one column per TDescriptor::Values() entry, in Descriptors() order
//   R    REAL, the double itself
//   VR   BLOB, msgpack array of float64          (per-frame series of a scalar; per-band statistics)
//   VVR  BLOB, msgpack array of arrays of float64 ([frame][band] series)
//
// (low-level descriptors carry kAllowBinaryStorage and not kAllowFloatingPointPrecisionStorage,
// SampleDescriptors.cpp:16-27, so every number is a float64: 0xcb + 8 bytes big-endian, msgpack-c 2.1.)
//
// LowLevelSchema() lists every column of the reference's low-level `assets` table with its sqlite type;
// LowLevelColumns() holds the values of the descriptors this library computes (everything but the file_*
// properties, which the caller knows, and rhythm_*).  SqlitePool.h writes them into a sqlite file.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "SampleAnalyser.h"

namespace afec {

struct TColumn {
  enum TType { kReal, kBlob };
  std::string mName;   // e.g. "spectral_centroid_VR", "cepstrum_bands_median_VR", "effectve_length_48dB_R"
  TType mType;
  double mReal;                 // kReal
  std::vector<uint8_t> mBlob;   // kBlob: msgpack
};

// name and declared sqlite type ("TEXT", "INTEGER", "REAL", "BLOB") of every descriptor column of the reference's
// low-level `assets` table, in its order (SqliteSampleDescriptorPool.cpp:1313-1358 on
// TSampleDescriptors::Descriptors(kLowLevelDescriptors)); filename / modtime / status precede them
struct TColumnSpec {
  std::string mName;
  const char* mpSqliteType;
};
std::vector<TColumnSpec> LowLevelSchema();

// msgpack encodings of SToMsgpack (SqliteSampleDescriptorPool.cpp:596-713)
std::vector<uint8_t> ToMsgpack(const double* pValues, size_t Count);
std::vector<uint8_t> ToMsgpack(const double* pValues, size_t Rows, size_t Width);   // [Rows][Width]

// the columns of one analysed sample, in the reference's column order (SampleDescriptors.cpp:150-205);
// pInfo adds analyzation_offset_R (SampleAnalyser.cpp:748-749)
std::vector<TColumn> LowLevelColumns(const TSampleDescriptors& Descriptors, const TSampleDataInfo* pInfo = nullptr,
                                     int SampleRate = 44100);
// the same into a vector kept between files: the names are built on the first call only, later calls overwrite the
// values and re-pack the BLOBs in place (what the database writer's per-file cost consists of)
void RefillLowLevelColumns(std::vector<TColumn>& Columns, const TSampleDescriptors& Descriptors,
                           const TSampleDataInfo* pInfo = nullptr, int SampleRate = 44100);

}  // namespace afec
