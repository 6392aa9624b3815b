// host_core.c — the marshalling core of the compiled host (pywfa_amd/host/_host.pyx): from the callers' str / bytes objects to the
// one upper-cased ASCII blob + offsets + lengths the C ABI takes (include/wfa_hip.h: wfa_hip_align_batch, wfa_hip_batch_create).
// What pywfa does per call — text.upper().encode("ascii") (/root/reference/pywfa/align.pyx:432,435) — done for a whole list at
// once by OpenMP threads that READ the objects' buffers in place.  The caller holds the GIL for the whole call (no Python code can
// run, so no object moves or dies); the worker threads touch no reference count and call no Python API.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>

// the bytes of an exact str (compact ASCII) or bytes object; 0 on anything else (the slow Python path converts those and raises
// the reference's errors: UnicodeEncodeError for non-ASCII text, AttributeError for other types)
static inline int view_of(PyObject* o, const uint8_t** data, Py_ssize_t* n, int* is_str) {
  if (PyUnicode_CheckExact(o)) {
    if (!PyUnicode_IS_READY(o) || !PyUnicode_IS_COMPACT_ASCII(o)) return 0;
    *data = (const uint8_t*)PyUnicode_DATA(o); *n = PyUnicode_GET_LENGTH(o); *is_str = 1;
    return 1;
  }
  if (PyBytes_CheckExact(o)) {
    *data = (const uint8_t*)PyBytes_AS_STRING(o); *n = PyBytes_GET_SIZE(o); *is_str = 0;
    return 1;
  }
  return 0;
}

// pass 1: lengths of the pairs' sequences (interleaved pattern, text, pattern, text ... when `patterns` is given, texts only
// otherwise) into len64[]; returns the number of objects the fast path cannot take (0 = go on)
int64_t wfa_host_lengths(PyObject** patterns, PyObject** texts, int64_t n, int64_t* len64, int threads) {
  int64_t bad = 0;
  const int step = patterns ? 2 : 1;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(+ : bad) if (n >= 4096)
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* d; Py_ssize_t ln; int is_str;
    if (patterns) {
      if (view_of(patterns[i], &d, &ln, &is_str) && ln <= 0x7ffffff0) len64[2 * i] = ln; else { len64[2 * i] = 0; ++bad; }
    }
    if (view_of(texts[i], &d, &ln, &is_str) && ln <= 0x7ffffff0) len64[step * i + step - 1] = ln; else { len64[step * i + step - 1] = 0; ++bad; }
  }
  return bad;
}

static inline void copy_upper(uint8_t* dst, const uint8_t* src, Py_ssize_t n) {
  for (Py_ssize_t i = 0; i < n; ++i) { const uint8_t c = src[i]; dst[i] = (uint8_t)(c - (((uint8_t)(c - 97) < 26) << 5)); }
}

// pass 2: the bytes into the blob at the offsets pass 1's prefix sum gave (str: upper-cased; bytes: as they are)
void wfa_host_fill(PyObject** patterns, PyObject** texts, int64_t n, const int64_t* p_off, const int64_t* t_off, uint8_t* blob, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static) if (n >= 4096)
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* d; Py_ssize_t ln; int is_str;
    if (patterns && view_of(patterns[i], &d, &ln, &is_str)) { if (is_str) copy_upper(blob + p_off[i], d, ln); else memcpy(blob + p_off[i], d, (size_t)ln); }
    if (view_of(texts[i], &d, &ln, &is_str)) { if (is_str) copy_upper(blob + t_off[i], d, ln); else memcpy(blob + t_off[i], d, (size_t)ln); }
  }
}
