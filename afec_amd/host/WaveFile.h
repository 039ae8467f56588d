// afec_amd/host/WaveFile.h -- RIFF / WAVE reader in front of the GPU LoadSample front end (first
// clause): the read side of the reference's TWaveFile (Source/Core/CoreFileFormats/Export/WaveFile.h,
// Source/WaveFile.cpp:365-410; chunk walk: Source/RiffFile.cpp:176-228).  Same acceptance rules and the same
// error texts ("Not a valid WAV file.", "Unsupported file format.", "Unsupported file format or corrupt file."),
// so that a crawl marks the same files failed (UnitTests.cpp:338-350: "_Not A Wavefile.wav").
//
// No sample is converted here: the data chunk is handed to afx_batch_create_from_raw as it lies in the file
// (int16 / packed int24 / int32 / float32 / float64, little endian); only 8-bit unsigned PCM is widened to
// int16, (v - 128) << 8, which is exactly what S8BitUnsignedTo16BitFloat reads (SampleConverter.h:392-395).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "SampleAnalyser.h"

namespace afec {

class TWaveFile {
public:
  // TAudioFile::TSampleType (Export/AudioFile.h) restricted to what a WAV can hold
  enum TSampleType { kInvalidSampleType = -1, k8BitUnsigned, k16Bit, k24Bit, k32BitInt, k32BitFloat, k64BitFloat };

  TWaveFile() = default;
  ~TWaveFile();
  TWaveFile(const TWaveFile&) = delete;
  TWaveFile& operator=(const TWaveFile&) = delete;

  // TAudioFile::OpenForRead: parses the chunks of a file on disk / of a file image in memory (the image must
  // outlive this object); throws TReadableException with the reference's texts.  Of a file on disk only the head
  // is read here; the file stays open until Close() or the destructor.
  void OpenForRead(const std::string& FileName);
  void OpenForRead(const void* pImage, size_t SizeInBytes, const std::string& Name = "<memory>");
  void Close();

  int NumChannels() const { return mChannels; }
  int SamplingRate() const { return mSampleRate; }
  int BitsPerSample() const { return mBitsPerSample; }
  TSampleType SampleType() const { return mSampleType; }
  int64_t NumSamples() const { return mNumOfSamples; }     // sample frames (TAudioFile::NumSamples)
  size_t FileSizeInBytes() const { return mSize; }

  // the data chunk as the GPU front end takes it: points into the file image (or into Storage, which receives
  // the samples of a file on disk and the widened samples of an 8-bit file)
  TDecodedSample DecodedSample(std::vector<unsigned char>& Storage) const;
  // The same in two steps, for callers that own the destination (the crawler's page-locked staging buffer):
  // format / channels / length without a pointer, then SampleDataBytes() bytes written to pDst -- for a file on
  // disk by pread, straight from the page cache.
  TDecodedSample DescribeSample() const;
  size_t SampleDataBytes() const;
  void ReadSampleData(void* pDst) const;

private:
  void Parse();
  bool ReadAt(size_t Position, void* pDst, size_t Bytes) const;

  static constexpr size_t kPrefixBytes = 4096;
  std::vector<unsigned char> mOwned;   // the first kPrefixBytes of a file opened by name
  int mFd = -1;                        // a file opened by name
  const unsigned char* mpImage = nullptr;
  size_t mImageBytes = 0;              // bytes behind mpImage: the whole file (images) or its prefix
  size_t mSize = 0;                    // of the file
  std::string mName;
  int mChannels = 0, mSampleRate = 0, mBitsPerSample = 0;
  TSampleType mSampleType = kInvalidSampleType;
  int64_t mNumOfSamples = 0;
  size_t mDataOffset = 0;
};

}  // namespace afec
