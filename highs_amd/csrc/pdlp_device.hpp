// pdlp_device.hpp — HIP error check + RAII device array.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <stdexcept>
#include <string>

namespace pdlp {

#define PDLP_HIP(expr)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e_) + " at " __FILE__ \
                               ":" + std::to_string(__LINE__) + " in " #expr);                      \
  } while (0)

template <typename T>
class DeviceArray {
 public:
  DeviceArray() = default;
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  DeviceArray(DeviceArray&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DeviceArray& operator=(DeviceArray&& o) noexcept {
    if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
    return *this;
  }
  ~DeviceArray() { release(); }
  void alloc(size_t count) {
    release();
    n_ = count;
    PDLP_HIP(hipMalloc(&p_, sizeof(T) * (count ? count : 1)));
  }
  void release() {
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    n_ = 0;
  }
  void upload(const T* host, size_t count, hipStream_t s) {
    if (count) PDLP_HIP(hipMemcpyAsync(p_, host, sizeof(T) * count, hipMemcpyHostToDevice, s));
  }
  void download(T* host, size_t count, hipStream_t s) const {
    if (count) PDLP_HIP(hipMemcpyAsync(host, p_, sizeof(T) * count, hipMemcpyDeviceToHost, s));
  }
  void zero(hipStream_t s) {
    if (n_) PDLP_HIP(hipMemsetAsync(p_, 0, sizeof(T) * n_, s));
  }
  T* get() const { return p_; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

}  // namespace pdlp
