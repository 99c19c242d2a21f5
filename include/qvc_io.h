/*
 * qvc_io.h -- C ABI of the host-side file I/O for corpus-scale conversion (libqvc_io.so; plain C++, no GPU code).
 *
 * The on-disk formats either side of the hot path are the reference's own:
 *   units  : one .npy per utterance, a C-ordered little-endian float32 array of shape (frames, 256), as written by
 *            dataset/encode.py:33-38 (np.save) and read back by data_utils_new_new.py:121-122;
 *   output : one mono float32 .wav per utterance at data.sampling_rate, as written by convert.py:84-86
 *            (scipy.io.wavfile.write(path, rate, float32 array)) -- the writer here produces the same bytes.
 * At ~2 ms of GPU time per 32 utterances the reference's per-line np.load / wavfile.write loop (convert.py:58-86) would
 * be the whole run time; this library reads a batch of unit files straight into a (pinned) batch buffer and writes a
 * batch of waveforms from one, on a pool of worker threads, while the caller's thread keeps the GPU fed.
 *
 * Conventions as in qvc.h: extern "C", plain pointers and sizes, int status (0 = ok, negative = error), nothing throws.
 */
#ifndef QVC_IO_H
#define QVC_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  QVC_IO_OK = 0,
  QVC_IO_ERR_BAD_ARG = -1,
  QVC_IO_ERR_OPEN = -2,      /* a file could not be opened / created              */
  QVC_IO_ERR_FORMAT = -3,    /* not a little-endian float32 C-ordered 2-D .npy    */
  QVC_IO_ERR_SHAPE = -4,     /* wrong column count, or more frames than the slot  */
  QVC_IO_ERR_IO = -5         /* short read / write                                 */
};

typedef struct qvc_io_pool qvc_io_pool;

/* A pool of `threads` worker threads (>= 1); calls below that take a pool run their per-file work on it and return
 * when all of it is done.  Several host threads may use one pool at the same time. */
int qvc_io_pool_create(int32_t threads, qvc_io_pool** out);
int qvc_io_pool_destroy(qvc_io_pool* pool);

/* Shape of a unit file from its header alone (no payload is read): dataset/encode.py:38 writes (frames, 256). */
int qvc_io_npy_shape(const char* path, int32_t* frames, int32_t* cols);

/* The same for n files at once, on the pool (planning a corpus run reads every header once). */
int qvc_io_npy_shapes(qvc_io_pool* pool, const char* const* paths, int32_t n, int32_t* frames, int32_t* cols);

/* Read n unit files into a batch buffer, FRAME-MAJOR as they are on disk: file i's (frames_i, cols) array lands at
 * dst + i * slot_frames * cols (row pitch = cols floats); rows past frames_i are left untouched (the ragged kernels
 * never read them).  frames_out[i] = frames_i.  Fails with QVC_IO_ERR_SHAPE if a file has another column count or
 * more than slot_frames frames.  Returns the first error of any file. */
int qvc_io_load_units(qvc_io_pool* pool, const char* const* paths, int32_t n, float* dst, int32_t slot_frames,
                      int32_t cols, int32_t* frames_out);

/* Write n mono float32 wav files: file i holds samples[i] floats from src + i * stride -- byte for byte what
 * scipy.io.wavfile.write(paths[i], rate, that float32 array) writes (convert.py:84-86). */
int qvc_io_write_wavs(qvc_io_pool* pool, const char* const* paths, int32_t n, const float* src, int64_t stride,
                      const int32_t* samples, int32_t rate);

#ifdef __cplusplus
}
#endif
#endif
