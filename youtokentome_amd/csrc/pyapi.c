/* pyapi.c -- the Python list boundary of encode(), in C: what yttm.pyx:87-109 does in Cython for the reference.
 *
 *   encode_ids(fn_encode, fn_free, handle, sentences, bos, eos, reverse, dropout_prob) -> list[list[int]]
 *
 * list/tuple of str -> one UTF-8 blob + offsets (PyUnicode_AsUTF8AndSize: an ASCII str IS its UTF-8) -> yttm_encode_as_ids
 * through the function pointer handed over by the ctypes loader (this module links against nothing but libpython) ->
 * list[list[int]] built straight from ids / out_offsets.  The int objects come from a table of the ids seen so far (token ids
 * are small and repeat: a cached object costs an INCREF instead of an allocation).  The GPU call runs with the GIL released.
 * Plain C (gcc), CPython C API; built by youtokentome_amd/csrc/Makefile into youtokentome_amd/_yttm_pyapi.so. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int (*encode_fn)(void *enc, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_sent, int bos, int eos, int reverse,
                         double dropout_prob, int32_t **ids, uint64_t **out_offsets, char *err, int errlen);
typedef void (*free_fn)(void *p);

static PyObject **g_int_cache = NULL; /* g_int_cache[i] = the int object i, for 0 <= i < g_int_cache_n */
static Py_ssize_t g_int_cache_n = 0;

static int grow_int_cache(Py_ssize_t need) {
  if (need <= g_int_cache_n) return 0;
  Py_ssize_t n = g_int_cache_n ? g_int_cache_n : 1024;
  while (n < need) n *= 2;
  PyObject **p = (PyObject **)realloc(g_int_cache, (size_t)n * sizeof(PyObject *));
  if (!p) {
    PyErr_NoMemory();
    return -1;
  }
  g_int_cache = p;
  for (Py_ssize_t i = g_int_cache_n; i < n; i++) {
    g_int_cache[i] = PyLong_FromSsize_t(i);
    if (!g_int_cache[i]) {
      g_int_cache_n = i;
      return -1;
    }
  }
  g_int_cache_n = n;
  return 0;
}

static PyObject *encode_ids(PyObject *self, PyObject *args) {
  (void)self;
  unsigned long long fn_enc_u, fn_free_u, handle_u;
  PyObject *sentences;
  int bos, eos, reverse;
  double dropout;
  if (!PyArg_ParseTuple(args, "KKKOpppd", &fn_enc_u, &fn_free_u, &handle_u, &sentences, &bos, &eos, &reverse, &dropout)) return NULL;
  encode_fn fn_enc = (encode_fn)(uintptr_t)fn_enc_u;
  free_fn fn_free = (free_fn)(uintptr_t)fn_free_u;
  PyObject *seq = PySequence_Fast(sentences, "sentences must be a list or a tuple of str");
  if (!seq) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  PyObject **items = PySequence_Fast_ITEMS(seq);
  uint64_t *offs = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
  if (!offs) {
    Py_DECREF(seq);
    return PyErr_NoMemory();
  }
  /* pass 1: sizes (the UTF-8 form is cached in the str object, pass 2 gets it for free) */
  uint64_t total = 0;
  offs[0] = 0;
  for (Py_ssize_t i = 0; i < n; i++) {
    Py_ssize_t len;
    if (!PyUnicode_Check(items[i])) {
      PyErr_Format(PyExc_TypeError, "sentence %zd is not a str", i);
      free(offs);
      Py_DECREF(seq);
      return NULL;
    }
    if (!PyUnicode_AsUTF8AndSize(items[i], &len)) {
      free(offs);
      Py_DECREF(seq);
      return NULL;
    }
    total += (uint64_t)len;
    offs[i + 1] = total;
  }
  uint8_t *blob = (uint8_t *)malloc(total ? total : 1);
  if (!blob) {
    free(offs);
    Py_DECREF(seq);
    return PyErr_NoMemory();
  }
  for (Py_ssize_t i = 0; i < n; i++) {
    Py_ssize_t len;
    const char *u = PyUnicode_AsUTF8AndSize(items[i], &len);
    memcpy(blob + offs[i], u, (size_t)len);
  }
  Py_DECREF(seq);
  int32_t *ids = NULL;
  uint64_t *out_off = NULL;
  char err[2048];
  err[0] = 0;
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = fn_enc((void *)(uintptr_t)handle_u, blob, offs, (uint64_t)n, bos, eos, reverse, dropout, &ids, &out_off, err, (int)sizeof err);
  Py_END_ALLOW_THREADS
  free(blob);
  free(offs);
  if (rc != 0) {
    PyErr_SetString(PyExc_ValueError, err); /* yttm.pyx: Status -> ValueError(message) */
    return NULL;
  }
  /* A million fresh lists make the cyclic collector run generation after generation over objects that cannot form a cycle (lists
   * of ints): with it on, building the result costs 1.0 s per 10^6 sentences, with it off 0.24 s.  It is switched off for the loop only
   * and put back the way it was. */
#if PY_VERSION_HEX >= 0x030A0000
  const int gc_was_on = PyGC_Disable();
#endif
  PyObject *out = PyList_New(n);
  if (!out) goto fail;
  for (Py_ssize_t i = 0; i < n; i++) {
    const uint64_t a = out_off[i], b = out_off[i + 1];
    PyObject *row = PyList_New((Py_ssize_t)(b - a));
    if (!row) goto fail;
    PyList_SET_ITEM(out, i, row);
    for (uint64_t j = a; j < b; j++) {
      const int32_t v = ids[j];
      PyObject *o;
      if (v >= 0 && v < (1 << 22)) { /* (every real token id; anything else gets a fresh object) */
        if (v >= g_int_cache_n && grow_int_cache((Py_ssize_t)v + 1) != 0) goto fail;
        o = g_int_cache[v];
        Py_INCREF(o);
      } else {
        o = PyLong_FromLong((long)v);
        if (!o) goto fail;
      }
      PyList_SET_ITEM(row, (Py_ssize_t)(j - a), o);
    }
  }
#if PY_VERSION_HEX >= 0x030A0000
  if (gc_was_on) PyGC_Enable();
#endif
  fn_free(ids);
  fn_free(out_off);
  return out;
fail:
#if PY_VERSION_HEX >= 0x030A0000
  if (gc_was_on) PyGC_Enable();
#endif
  Py_XDECREF(out);
  fn_free(ids);
  fn_free(out_off);
  return NULL;
}

static PyMethodDef methods[] = {
    {"encode_ids", encode_ids, METH_VARARGS, "encode_ids(fn_encode, fn_free, handle, sentences, bos, eos, reverse, dropout_prob) -> list[list[int]]"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_yttm_pyapi", "list[str] <-> C ABI marshalling for youtokentome_amd (yttm.pyx:87-109)", -1, methods,
                                    NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__yttm_pyapi(void) { return PyModule_Create(&moddef); }
