// afec_amd/host/WaveFile.cpp -- see WaveFile.h.
#include "WaveFile.h"

#include <cstdio>
#include <cstring>

#include "../../include/afx.h"

namespace afec {

namespace {

constexpr uint16_t kWaveFormatPcm = 1, kWaveFormatIeeeFloat = 3, kWaveFormatExtensible = 0xFFFE;

uint16_t Read16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t Read32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct TChunk {
  char mName[5];
  size_t mOffset;   // of the chunk's data
  uint32_t mSize;
};

}  // namespace

void TWaveFile::OpenForRead(const std::string& FileName) {
  FILE* pFile = std::fopen(FileName.c_str(), "rb");
  if (!pFile) throw TReadableException("Failed to open the file '" + FileName + "'.");   // RiffFile.cpp:128-131
  std::fseek(pFile, 0, SEEK_END);
  const long Size = std::ftell(pFile);
  std::fseek(pFile, 0, SEEK_SET);
  mOwned.resize(Size > 0 ? (size_t)Size : 0);
  const size_t Got = mOwned.empty() ? 0 : std::fread(mOwned.data(), 1, mOwned.size(), pFile);
  std::fclose(pFile);
  if (Got != mOwned.size()) throw TReadableException("Failed to open the file '" + FileName + "'.");
  mpImage = mOwned.data();
  mSize = mOwned.size();
  mName = FileName;
  Parse();
}

void TWaveFile::OpenForRead(const void* pImage, size_t SizeInBytes, const std::string& Name) {
  mOwned.clear();
  mpImage = static_cast<const unsigned char*>(pImage);
  mSize = SizeInBytes;
  mName = Name;
  Parse();
}

void TWaveFile::Parse() {
  // TRiffFile::ReadChunks (RiffFile.cpp:176-228): skip the 8-byte RIFF header, then walk chunk headers; "WAVE"
  // is a parent chunk (a name without size and data), every other chunk is word aligned; the walk stops at the
  // first header that does not fit.  Like the reference, the "RIFF" tag itself is not checked: a file is a WAV
  // when the walk finds "WAVE", "fmt " and "data".
  std::vector<TChunk> Chunks;
  size_t Position = 8;
  while (Position + 8 <= mSize) {
    TChunk c;
    std::memcpy(c.mName, mpImage + Position, 4);
    c.mName[4] = 0;
    c.mSize = Read32(mpImage + Position + 4);
    c.mOffset = Position + 8;
    Chunks.push_back(c);
    if (!std::strcmp(c.mName, "WAVE")) {
      Position += 4;                       // parent chunks have only a name
      continue;
    }
    const size_t Next = c.mOffset + c.mSize + (c.mSize & 1);
    if (Next > c.mOffset && Next + 8 < mSize) Position = Next;   // RiffFile.cpp:206: against the position behind the header, so an empty chunk ends the walk
    else break;
  }
  auto Find = [&](const char* pName) -> const TChunk* {
    for (const TChunk& c : Chunks)
      if (!std::strcmp(c.mName, pName)) return &c;
    return nullptr;
  };
  const TChunk* pFormat = Find("fmt ");
  const TChunk* pData = Find("data");
  if (!Find("WAVE") || !pData || !pFormat) throw TReadableException("Not a valid WAV file.");   // WaveFile.cpp:378-383

  // TWaveFormatChunkData::Read (WaveFile.cpp:87-95) + TWaveFormatChunk::VerifyValidity (:127-152)
  if (pFormat->mOffset + 16 > mSize) throw TReadableException("Unsupported file format.");
  const unsigned char* f = mpImage + pFormat->mOffset;
  const uint16_t FormatTag = Read16(f), Channels = Read16(f + 2), BitsPerSample = Read16(f + 14);
  const uint32_t SampleRate = Read32(f + 4), AvgBytesPerSec = Read32(f + 8);
  const bool TagOk = FormatTag == kWaveFormatPcm || FormatTag == kWaveFormatIeeeFloat || FormatTag == kWaveFormatExtensible;
  const bool BitsOk = BitsPerSample == 8 || BitsPerSample == 16 || BitsPerSample == 24 || BitsPerSample == 32 || BitsPerSample == 64;
  if (!TagOk || !BitsOk || AvgBytesPerSec != (uint32_t)((uint64_t)Channels * SampleRate * BitsPerSample / 8))
    throw TReadableException("Unsupported file format.");
  if (SampleRate == 0) throw TReadableException("Unsupported file format.");   // (the byte-rate check passes for 0 == 0)

  // WaveFile.cpp:398-405
  const uint32_t FrameBytes = (uint32_t)Channels * (BitsPerSample / 8);
  if (FrameBytes == 0) throw TReadableException("Unsupported file format or corrupt file.");
  // a data chunk that claims more than the file holds (truncated copies) is cut to what is there
  const size_t Available = mSize > pData->mOffset ? mSize - pData->mOffset : 0;
  const size_t DataBytes = pData->mSize <= Available ? pData->mSize : Available;
  mNumOfSamples = (int64_t)(DataBytes / FrameBytes);
  if (mNumOfSamples <= 0) throw TReadableException("Unsupported file format or corrupt file.");

  mChannels = Channels;
  mSampleRate = (int)SampleRate;
  mBitsPerSample = BitsPerSample;
  mDataOffset = pData->mOffset;
  // TWaveFormatChunkData::SSampleType (WaveFile.cpp:14-47)
  switch (BitsPerSample) {
    case 8: mSampleType = k8BitUnsigned; break;
    case 16: mSampleType = k16Bit; break;
    case 24: mSampleType = k24Bit; break;
    case 32: mSampleType = (FormatTag == kWaveFormatExtensible || FormatTag == kWaveFormatIeeeFloat) ? k32BitFloat : k32BitInt; break;
    default: mSampleType = k64BitFloat; break;
  }
}

TDecodedSample TWaveFile::DecodedSample(std::vector<unsigned char>& Storage) const {
  if (!mpImage || mNumOfSamples <= 0) throw TReadableException("TWaveFile: no file is open");
  TDecodedSample s;
  s.mNumberOfChannels = mChannels;
  s.mSampleRate = mSampleRate;
  s.mNumberOfSampleFrames = mNumOfSamples;
  s.mpInterleavedSamples = mpImage + mDataOffset;
  switch (mSampleType) {
    case k16Bit: s.mFormat = AFX_RAW_I16; break;
    case k24Bit: s.mFormat = AFX_RAW_I24; break;
    case k32BitInt: s.mFormat = AFX_RAW_I32; break;
    case k32BitFloat: s.mFormat = AFX_RAW_F32; break;
    case k64BitFloat: s.mFormat = AFX_RAW_F64; break;
    default: {
      // 8-bit unsigned: (v - 128) << 8 as int16 (S8BitUnsignedTo16BitFloat, SampleConverter.h:392-395)
      const size_t n = (size_t)mNumOfSamples * mChannels;
      Storage.resize(n * 2);
      int16_t* pDst = reinterpret_cast<int16_t*>(Storage.data());
      const unsigned char* pSrc = mpImage + mDataOffset;
      for (size_t i = 0; i < n; ++i) pDst[i] = (int16_t)(((int)pSrc[i] - 128) * 256);
      s.mFormat = AFX_RAW_I16;
      s.mpInterleavedSamples = Storage.data();
      break;
    }
  }
  return s;
}

}  // namespace afec
