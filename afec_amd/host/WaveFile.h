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

  // TAudioFile::OpenForRead: parses the chunks of a file on disk / of a file image in memory (the image must
  // outlive this object); throws TReadableException with the reference's texts
  void OpenForRead(const std::string& FileName);
  void OpenForRead(const void* pImage, size_t SizeInBytes, const std::string& Name = "<memory>");

  int NumChannels() const { return mChannels; }
  int SamplingRate() const { return mSampleRate; }
  int BitsPerSample() const { return mBitsPerSample; }
  TSampleType SampleType() const { return mSampleType; }
  int64_t NumSamples() const { return mNumOfSamples; }     // sample frames (TAudioFile::NumSamples)
  size_t FileSizeInBytes() const { return mSize; }

  // the data chunk as the GPU front end takes it: points into the file image (or, for 8-bit files, into
  // Storage, which receives the widened samples)
  TDecodedSample DecodedSample(std::vector<unsigned char>& Storage) const;

private:
  void Parse();

  std::vector<unsigned char> mOwned;   // file contents when opened by name
  const unsigned char* mpImage = nullptr;
  size_t mSize = 0;
  std::string mName;
  int mChannels = 0, mSampleRate = 0, mBitsPerSample = 0;
  TSampleType mSampleType = kInvalidSampleType;
  int64_t mNumOfSamples = 0;
  size_t mDataOffset = 0;
};

}  // namespace afec
