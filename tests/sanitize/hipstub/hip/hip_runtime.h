// tests/sanitize/hipstub/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY (never on an include path of afec_amd/): the part
// of the HIP runtime API that the C-ABI's host code (afec_amd/csrc/afx_*.cpp) uses, backed by host memory
// (hip_stub.cpp), so that this code can be compiled by g++ with -fsanitize=address,undefined / thread and driven on a
// machine without a GPU.  "Device memory" is the host heap: a table upload that overruns its buffer, a kernel argument
// that points past an allocation, a record the planner sized too small are heap-buffer-overflows the sanitizer reports.
// The kernels' launchers (afx_internal.h) are mock_kernels.cpp: they walk exactly the tables a kernel would and assert
// the invariants the kernels rely on.
#pragma once

#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNotReady = 600,
  hipErrorNoDevice = 100,
  hipErrorInvalidDevice = 101,
  hipErrorUnknown = 999
} hipError_t;

typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;

enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };
enum { hipHostRegisterDefault = 0, hipHostRegisterPortable = 1 };
enum { hipDeviceScheduleBlockingSync = 4 };
typedef enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 } hipDeviceAttribute_t;

struct float2 { float x, y; };
struct double2 { double x, y; };

typedef struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  size_t totalGlobalMem;
  int multiProcessorCount;
} hipDeviceProp_t;

extern "C" {
hipError_t hipGetDeviceCount(int* count);
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int* device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device);
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int device);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t stream);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
}
template <typename K>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* blocks, K, int, size_t) { *blocks = 1; return hipSuccess; }

// ---- the mock device's own controls (tests only) ----
namespace hipstub {
void set_device_count(int n);                 // devices the "runtime" reports (default 1)
void set_memory_limit(size_t bytes);          // device bytes that may be allocated at once on one device (0: unlimited)
void fail_allocation_after(long n);           // the n-th hipMalloc from now fails with hipErrorOutOfMemory once (< 0: never)
void lose_device(int device, bool lost);      // every call on a lost device returns hipErrorUnknown
size_t device_bytes_in_use();                 // all devices
long live_streams();
long live_events();
}  // namespace hipstub
