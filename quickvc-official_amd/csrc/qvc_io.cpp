// qvc_io.cpp -- host-side batch file I/O for corpus-scale conversion (include/qvc_io.h): unit .npy files in, float32
// wav files out, on a pool of worker threads.  Plain C++17 + POSIX; no GPU code, no dependency on the HIP library.
#include "../../include/qvc_io.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct qvc_io_pool {
  std::vector<std::thread> workers;
  std::deque<std::function<void()>> jobs;
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;

  explicit qvc_io_pool(int n) {
    for (int i = 0; i < n; ++i)
      workers.emplace_back([this] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return stop || !jobs.empty(); });
            if (jobs.empty()) return;                 // stop requested and nothing left
            job = std::move(jobs.front());
            jobs.pop_front();
          }
          job();
        }
      });
  }
  ~qvc_io_pool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto& t : workers) t.join();
  }
  // run f(0..n-1) on the pool, return when all are done; first non-zero result wins
  int run(int n, const std::function<int(int)>& f) {
    if (n <= 0) return QVC_IO_OK;
    struct Group { std::mutex m; std::condition_variable c; int left; std::atomic<int> err{0}; } g;
    g.left = n;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (int i = 0; i < n; ++i)
        jobs.emplace_back([&g, &f, i] {
          const int rc = f(i);
          if (rc != 0) { int zero = 0; g.err.compare_exchange_strong(zero, rc); }
          std::lock_guard<std::mutex> l2(g.m);
          if (--g.left == 0) g.c.notify_all();
        });
    }
    cv.notify_all();
    std::unique_lock<std::mutex> lk(g.m);
    g.c.wait(lk, [&g] { return g.left == 0; });
    return g.err.load();
  }
};

namespace {

struct NpyInfo { int64_t rows = 0, cols = 0, data_off = 0; };

// Header of a .npy file (format 1.0 / 2.0 / 3.0): magic, version, little-endian header length, then a Python dict
// literal {'descr': '<f4', 'fortran_order': False, 'shape': (rows, cols), }.
int read_npy_header(int fd, NpyInfo& out) {
  unsigned char pre[12];
  if (pread(fd, pre, 10, 0) != 10 || std::memcmp(pre, "\x93NUMPY", 6) != 0) return QVC_IO_ERR_FORMAT;
  const int major = pre[6];
  int64_t hlen, hoff;
  if (major == 1) { hlen = pre[8] | (pre[9] << 8); hoff = 10; }
  else if (major == 2 || major == 3) {
    if (pread(fd, pre, 12, 0) != 12) return QVC_IO_ERR_FORMAT;
    hlen = (int64_t)pre[8] | ((int64_t)pre[9] << 8) | ((int64_t)pre[10] << 16) | ((int64_t)pre[11] << 24);
    hoff = 12;
  } else return QVC_IO_ERR_FORMAT;
  if (hlen <= 0 || hlen > 65536) return QVC_IO_ERR_FORMAT;
  std::string h((size_t)hlen, '\0');
  if (pread(fd, &h[0], (size_t)hlen, hoff) != hlen) return QVC_IO_ERR_FORMAT;
  auto value_after = [&](const char* key) -> size_t {
    const size_t k = h.find(key);
    if (k == std::string::npos) return std::string::npos;
    const size_t c = h.find(':', k);
    return c == std::string::npos ? c : c + 1;
  };
  size_t p = value_after("'descr'");
  if (p == std::string::npos) return QVC_IO_ERR_FORMAT;
  const size_t q0 = h.find('\'', p);
  const size_t q1 = q0 == std::string::npos ? q0 : h.find('\'', q0 + 1);
  if (q1 == std::string::npos) return QVC_IO_ERR_FORMAT;
  const std::string descr = h.substr(q0 + 1, q1 - q0 - 1);
  if (descr != "<f4" && descr != "=f4") return QVC_IO_ERR_FORMAT;             // little-endian float32 only
  p = value_after("'fortran_order'");
  if (p == std::string::npos || h.compare(h.find_first_not_of(' ', p), 5, "False") != 0) return QVC_IO_ERR_FORMAT;
  p = value_after("'shape'");
  if (p == std::string::npos) return QVC_IO_ERR_FORMAT;
  const size_t o = h.find('(', p), c = h.find(')', p);
  if (o == std::string::npos || c == std::string::npos || c < o) return QVC_IO_ERR_FORMAT;
  int64_t dims[3]; int nd = 0;
  const char* s = h.c_str() + o + 1;
  const char* end = h.c_str() + c;
  while (s < end && nd < 3) {
    while (s < end && (*s == ' ' || *s == ',')) ++s;
    if (s >= end) break;
    char* e = nullptr;
    const long long v = std::strtoll(s, &e, 10);
    if (e == s || v < 0) return QVC_IO_ERR_FORMAT;
    dims[nd++] = v; s = e;
  }
  if (nd != 2) return QVC_IO_ERR_FORMAT;
  out.rows = dims[0]; out.cols = dims[1]; out.data_off = hoff + hlen;
  return QVC_IO_OK;
}

int read_all(int fd, void* dst, size_t bytes, int64_t off) {
  char* p = static_cast<char*>(dst);
  while (bytes > 0) {
    const ssize_t n = pread(fd, p, bytes, off);
    if (n <= 0) return QVC_IO_ERR_IO;
    p += n; off += n; bytes -= (size_t)n;
  }
  return QVC_IO_OK;
}

void put32(unsigned char* p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
void put16(unsigned char* p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

// scipy.io.wavfile.write for a 1-D float32 array: RIFF / WAVE, an 18-byte fmt chunk (IEEE_FLOAT, cbSize 0), a fact
// chunk with the sample count, then the data chunk -- 58 bytes in front of the samples.
int write_wav(const char* path, const float* x, int32_t n, int32_t rate) {
  unsigned char h[58];
  const uint32_t bytes = (uint32_t)n * 4u;
  std::memcpy(h, "RIFF", 4); put32(h + 4, 50u + bytes); std::memcpy(h + 8, "WAVE", 4);
  std::memcpy(h + 12, "fmt ", 4); put32(h + 16, 18);
  put16(h + 20, 3); put16(h + 22, 1); put32(h + 24, (uint32_t)rate); put32(h + 28, (uint32_t)rate * 4u); put16(h + 32, 4); put16(h + 34, 32); put16(h + 36, 0);
  std::memcpy(h + 38, "fact", 4); put32(h + 42, 4); put32(h + 46, (uint32_t)n);
  std::memcpy(h + 50, "data", 4); put32(h + 54, bytes);
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
  if (fd < 0) return QVC_IO_ERR_OPEN;
  struct iovec iov[2] = {{h, sizeof(h)}, {const_cast<float*>(x), (size_t)bytes}};
  size_t left = sizeof(h) + bytes;
  int rc = QVC_IO_OK;
  int idx = 0;
  while (left > 0) {
    const ssize_t w = writev(fd, iov + idx, 2 - idx);
    if (w <= 0) { rc = QVC_IO_ERR_IO; break; }
    left -= (size_t)w;
    size_t adv = (size_t)w;
    while (idx < 2 && adv >= iov[idx].iov_len) { adv -= iov[idx].iov_len; ++idx; }
    if (idx < 2 && adv) { iov[idx].iov_base = static_cast<char*>(iov[idx].iov_base) + adv; iov[idx].iov_len -= adv; }
  }
  if (close(fd) != 0 && rc == QVC_IO_OK) rc = QVC_IO_ERR_IO;
  return rc;
}

}  // namespace

extern "C" {

int qvc_io_pool_create(int32_t threads, qvc_io_pool** out) {
  if (!out || threads < 1 || threads > 256) return QVC_IO_ERR_BAD_ARG;
  qvc_io_pool* p = new (std::nothrow) qvc_io_pool(threads);
  if (!p) return QVC_IO_ERR_BAD_ARG;
  *out = p;
  return QVC_IO_OK;
}

int qvc_io_pool_destroy(qvc_io_pool* pool) {
  if (!pool) return QVC_IO_ERR_BAD_ARG;
  delete pool;
  return QVC_IO_OK;
}

int qvc_io_npy_shape(const char* path, int32_t* frames, int32_t* cols) {
  if (!path || !frames || !cols) return QVC_IO_ERR_BAD_ARG;
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return QVC_IO_ERR_OPEN;
  NpyInfo info;
  const int rc = read_npy_header(fd, info);
  close(fd);
  if (rc != QVC_IO_OK) return rc;
  if (info.rows > INT32_MAX || info.cols > INT32_MAX) return QVC_IO_ERR_SHAPE;
  *frames = (int32_t)info.rows; *cols = (int32_t)info.cols;
  return QVC_IO_OK;
}

int qvc_io_npy_shapes(qvc_io_pool* pool, const char* const* paths, int32_t n, int32_t* frames, int32_t* cols) {
  if (!pool || !paths || !frames || !cols || n < 0) return QVC_IO_ERR_BAD_ARG;
  // headers are tiny: hand each worker a run of files instead of one job per file
  const int per = 64, jobs = (n + per - 1) / per;
  return pool->run(jobs, [=](int j) -> int {
    int rc = QVC_IO_OK;
    for (int i = j * per; i < n && i < (j + 1) * per; ++i) {
      const int r = paths[i] ? qvc_io_npy_shape(paths[i], frames + i, cols + i) : QVC_IO_ERR_BAD_ARG;
      if (r != QVC_IO_OK && rc == QVC_IO_OK) rc = r;
    }
    return rc;
  });
}

int qvc_io_load_units(qvc_io_pool* pool, const char* const* paths, int32_t n, float* dst, int32_t slot_frames, int32_t cols,
                      int32_t* frames_out) {
  if (!pool || !paths || !dst || !frames_out || n < 0 || slot_frames <= 0 || cols <= 0) return QVC_IO_ERR_BAD_ARG;
  return pool->run(n, [=](int i) -> int {
    if (!paths[i]) return QVC_IO_ERR_BAD_ARG;
    const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
    if (fd < 0) return QVC_IO_ERR_OPEN;
    NpyInfo info;
    int rc = read_npy_header(fd, info);
    if (rc == QVC_IO_OK && (info.cols != cols || info.rows > slot_frames)) rc = QVC_IO_ERR_SHAPE;
    if (rc == QVC_IO_OK) {
      frames_out[i] = (int32_t)info.rows;
      rc = read_all(fd, dst + (size_t)i * slot_frames * cols, (size_t)info.rows * cols * 4, info.data_off);
    }
    close(fd);
    return rc;
  });
}

int qvc_io_write_wavs(qvc_io_pool* pool, const char* const* paths, int32_t n, const float* src, int64_t stride,
                      const int32_t* samples, int32_t rate) {
  if (!pool || !paths || !src || !samples || n < 0 || stride < 0 || rate <= 0) return QVC_IO_ERR_BAD_ARG;
  return pool->run(n, [=](int i) -> int {
    if (!paths[i] || samples[i] < 0) return QVC_IO_ERR_BAD_ARG;
    return write_wav(paths[i], src + (size_t)i * stride, samples[i], rate);
  });
}

}  // extern "C"
