// afec_amd/host/WaveFile.cpp -- see WaveFile.h.
#include "WaveFile.h"

#include <emmintrin.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>

#include "../../include/afx.h"

namespace afec {

namespace {

constexpr uint16_t kWaveFormatPcm = 1, kWaveFormatIeeeFloat = 3, kWaveFormatExtensible = 0xFFFE;

uint16_t Read16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t Read32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// A data chunk copied out of a file image is written once and next read by the GPU's DMA engine: non-temporal
// stores (no read-for-ownership of the destination lines, nothing of it left in the caches) -- half the CPU time
// of memcpy for the crawler's staging copies (profiles/r03/README.md).
void CopyStreaming(void* pDst, const void* pSrc, size_t Bytes) {
  char* d = static_cast<char*>(pDst);
  const char* s = static_cast<const char*>(pSrc);
  if (Bytes < 4096) {
    std::memcpy(d, s, Bytes);
    return;
  }
  const size_t Head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
  std::memcpy(d, s, Head);
  d += Head, s += Head, Bytes -= Head;
  for (; Bytes >= 64; d += 64, s += 64, Bytes -= 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 32)), e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 48));
    _mm_stream_si128(reinterpret_cast<__m128i*>(d), a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 48), e);
  }
  std::memcpy(d, s, Bytes);
  _mm_sfence();
}

struct TChunk {
  char mName[5];
  size_t mOffset;   // of the chunk's data
  uint32_t mSize;
};

}  // namespace

TWaveFile::~TWaveFile() { Close(); }

void TWaveFile::Close() {
  if (mFd >= 0) ::close(mFd);
  mFd = -1;
}

// pread until Bytes have arrived (a short count is the end of the file or an error)
bool TWaveFile::ReadAt(size_t Position, void* pDst, size_t Bytes) const {
  size_t Got = 0;
  while (Got < Bytes) {
    const ssize_t r = ::pread(mFd, (char*)pDst + Got, Bytes - Got, (off_t)(Position + Got));
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) return false;
    Got += (size_t)r;
  }
  return true;
}

// A file on disk is parsed from its first kPrefixBytes (chunk headers further back are read one by one); the data
// chunk is not read here: ReadSampleData moves it from the page cache straight to where the caller wants it (the
// crawler's page-locked staging buffer) -- one copy per file instead of file -> vector -> staging.
void TWaveFile::OpenForRead(const std::string& FileName) {
  Close();
  mFd = ::open(FileName.c_str(), O_RDONLY | O_CLOEXEC);
  struct stat St;
  if (mFd < 0 || ::fstat(mFd, &St) != 0 || !S_ISREG(St.st_mode)) {
    Close();
    throw TReadableException("Failed to open the file '" + FileName + "'.");   // RiffFile.cpp:128-131
  }
  mSize = (size_t)St.st_size;
  mOwned.resize(mSize < kPrefixBytes ? mSize : kPrefixBytes);
  if (!mOwned.empty() && !ReadAt(0, mOwned.data(), mOwned.size())) {
    Close();
    throw TReadableException("Failed to open the file '" + FileName + "'.");
  }
  mpImage = mOwned.data();
  mImageBytes = mOwned.size();
  mName = FileName;
  try {
    Parse();
  } catch (...) {
    Close();
    throw;
  }
}

void TWaveFile::OpenForRead(const void* pImage, size_t SizeInBytes, const std::string& Name) {
  Close();
  mOwned.clear();
  mpImage = static_cast<const unsigned char*>(pImage);
  mSize = mImageBytes = SizeInBytes;
  mName = Name;
  Parse();
}

void TWaveFile::Parse() {
  // TRiffFile::ReadChunks (RiffFile.cpp:176-228): skip the 8-byte RIFF header, then walk chunk headers; "WAVE"
  // is a parent chunk (a name without size and data), every other chunk is word aligned; the walk stops at the
  // first header that does not fit.  Like the reference, the "RIFF" tag itself is not checked: a file is a WAV
  // when the walk finds "WAVE", "fmt " and "data".
  // n bytes of the file at Position: from the image / the prefix, else (files on disk) read into Scratch
  unsigned char Scratch[16];
  auto Bytes = [&](size_t Position, size_t n) -> const unsigned char* {
    if (Position + n <= mImageBytes) return mpImage + Position;
    if (mFd < 0 || n > sizeof(Scratch) || !ReadAt(Position, Scratch, n)) throw TReadableException("Not a valid WAV file.");
    return Scratch;
  };
  std::vector<TChunk> Chunks;
  size_t Position = 8;
  while (Position + 8 <= mSize) {
    TChunk c;
    const unsigned char* pHeader = Bytes(Position, 8);
    std::memcpy(c.mName, pHeader, 4);
    c.mName[4] = 0;
    c.mSize = Read32(pHeader + 4);
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
  const unsigned char* f = Bytes(pFormat->mOffset, 16);
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

size_t TWaveFile::SampleDataBytes() const {
  const size_t n = (size_t)mNumOfSamples * (size_t)mChannels;
  return mSampleType == k8BitUnsigned ? n * 2 : n * (size_t)(mBitsPerSample / 8);
}

TDecodedSample TWaveFile::DescribeSample() const {
  if (!mpImage || mNumOfSamples <= 0) throw TReadableException("TWaveFile: no file is open");
  TDecodedSample s;
  s.mNumberOfChannels = mChannels;
  s.mSampleRate = mSampleRate;
  s.mNumberOfSampleFrames = mNumOfSamples;
  s.mpInterleavedSamples = nullptr;
  switch (mSampleType) {
    case k24Bit: s.mFormat = AFX_RAW_I24; break;
    case k32BitInt: s.mFormat = AFX_RAW_I32; break;
    case k32BitFloat: s.mFormat = AFX_RAW_F32; break;
    case k64BitFloat: s.mFormat = AFX_RAW_F64; break;
    default: s.mFormat = AFX_RAW_I16; break;   // 16-bit, and 8-bit widened
  }
  return s;
}

void TWaveFile::ReadSampleData(void* pDst) const {
  if (!mpImage || mNumOfSamples <= 0) throw TReadableException("TWaveFile: no file is open");
  const size_t Raw = (size_t)mNumOfSamples * (size_t)mChannels * (size_t)(mBitsPerSample / 8);
  // the file's bytes: from the image, or the part the prefix holds + the rest from the file
  auto Fetch = [&](void* pTo) {
    const size_t Have = mDataOffset < mImageBytes ? std::min(mImageBytes - mDataOffset, Raw) : 0;
    if (Have) CopyStreaming(pTo, mpImage + mDataOffset, Have);
    if (Have < Raw && (mFd < 0 || !ReadAt(mDataOffset + Have, (char*)pTo + Have, Raw - Have)))
      throw TReadableException("Failed to read the file '" + mName + "'.");
  };
  if (mSampleType != k8BitUnsigned) {
    Fetch(pDst);
    return;
  }
  // 8-bit unsigned: (v - 128) << 8 as int16 (S8BitUnsignedTo16BitFloat, SampleConverter.h:392-395)
  const unsigned char* pSrc;
  std::vector<unsigned char> Temp;
  if (mDataOffset + Raw <= mImageBytes) pSrc = mpImage + mDataOffset;
  else {
    Temp.resize(Raw);
    Fetch(Temp.data());
    pSrc = Temp.data();
  }
  int16_t* pOut = static_cast<int16_t*>(pDst);
  for (size_t i = 0; i < Raw; ++i) pOut[i] = (int16_t)(((int)pSrc[i] - 128) * 256);
}

TDecodedSample TWaveFile::DecodedSample(std::vector<unsigned char>& Storage) const {
  TDecodedSample s = DescribeSample();
  const size_t Raw = (size_t)mNumOfSamples * (size_t)mChannels * (size_t)(mBitsPerSample / 8);
  if (mSampleType != k8BitUnsigned && mDataOffset + Raw <= mImageBytes) {
    s.mpInterleavedSamples = mpImage + mDataOffset;   // a file image: no copy
    return s;
  }
  Storage.resize(SampleDataBytes());
  ReadSampleData(Storage.data());
  s.mpInterleavedSamples = Storage.data();
  return s;
}

}  // namespace afec
