/* _bagtable: the bookkeeping of a trajectory list for brov_upload_bags (engine.BagTable) without a Python-level loop.
 *
 * KoopmanEDMDc.fit_multi(X_list, U_list) is handed tens of thousands of small arrays (BASELINE config 3: 20 000 + 20 000); reading each
 * one's header from Python (dtype, layout, shape, address) costs ~0.7 us, 30 ms per call -- as much as moving the 1.6 GB they hold.
 * Through the buffer protocol the same facts cost ~60 ns per array.  Host-side plumbing only: no arithmetic lives here.
 *
 *   fill(seq, ncols, start, ptr_out, rows_out) -> int
 *     seq      : list / tuple of objects
 *     ptr_out  : writable buffer of len(seq) uint64   (host address of every conforming array)
 *     rows_out : writable buffer of len(seq) int64    (its number of rows)
 *   Walks seq[start:].  An item conforms if it exports a C-contiguous 2-D buffer of doubles with `ncols` columns.  Returns -1 when every
 *   item did, else the index of the first one that does not (the caller converts that one and calls again from there).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static PyObject* bt_fill(PyObject* self, PyObject* args) {
    PyObject* seq;
    Py_ssize_t ncols, start;
    Py_buffer pbuf, rbuf;
    (void)self;
    if (!PyArg_ParseTuple(args, "Onnw*w*", &seq, &ncols, &start, &pbuf, &rbuf)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "fill: a sequence of arrays is required");
    if (!fast) { PyBuffer_Release(&pbuf); PyBuffer_Release(&rbuf); return NULL; }
    const Py_ssize_t nb = PySequence_Fast_GET_SIZE(fast);
    long long bad = -1;
    if (pbuf.len < (Py_ssize_t)(nb * 8) || rbuf.len < (Py_ssize_t)(nb * 8) || start < 0) {
        Py_DECREF(fast); PyBuffer_Release(&pbuf); PyBuffer_Release(&rbuf);
        PyErr_SetString(PyExc_ValueError, "fill: output buffers too small");
        return NULL;
    }
    uint64_t* ptr = (uint64_t*)pbuf.buf;
    int64_t* rows = (int64_t*)rbuf.buf;
    PyObject** items = PySequence_Fast_ITEMS(fast);
    for (Py_ssize_t b = start; b < nb; ++b) {
        Py_buffer v;
        if (PyObject_GetBuffer(items[b], &v, PyBUF_STRIDES | PyBUF_FORMAT) != 0) { PyErr_Clear(); bad = b; break; }
        const char* f = v.format ? v.format : "B";
        if (*f == '@' || *f == '=' || *f == '<') ++f;                  /* native / little-endian doubles */
        const int ok = v.ndim == 2 && v.itemsize == 8 && f[0] == 'd' && f[1] == 0 && v.shape[1] == ncols && PyBuffer_IsContiguous(&v, 'C');
        if (ok) { ptr[b] = (uint64_t)(uintptr_t)v.buf; rows[b] = (int64_t)v.shape[0]; }
        PyBuffer_Release(&v);                                          /* (the caller keeps the objects alive for as long as it uses the addresses) */
        if (!ok) { bad = b; break; }
    }
    Py_DECREF(fast);
    PyBuffer_Release(&pbuf);
    PyBuffer_Release(&rbuf);
    return PyLong_FromLongLong(bad);
}

static PyMethodDef bt_methods[] = {
    {"fill", bt_fill, METH_VARARGS, "fill(seq, ncols, start, ptr_out, rows_out) -> -1 or the index of the first non-conforming item"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef bt_module = {PyModuleDef_HEAD_INIT, "_bagtable", "array-list bookkeeping for brov_upload_bags", -1, bt_methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__bagtable(void) { return PyModule_Create(&bt_module); }
