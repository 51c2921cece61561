/* mvosr_pyhelper.c — libmvosr_py.so: the one place where the host side needs the CPython C API.
 *
 * The reference's call surface hands the estimator one NumPy array per frame (/root/reference/src/main.py:102-113,
 * main_offline.py:52-55); the batch path packs thousands of them per chunk, and asking each array for its data pointer
 * from Python costs ~2 us apiece — more than all GPU stages of the frame together.  This helper walks the two lists
 * through the buffer protocol in C (loaded with ctypes.PyDLL: the GIL is held) and fills the pointer / size tables the
 * C packer of libmvosr.so (mvosr_pack_fill) takes.  No arithmetic, no GPU, no libmvosr dependency. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* Returns the number of frames (>= 0) when every frame is a C-contiguous float64 (n,3) / (n,2) pair with the same n;
 * -(i+1) for the first frame i that is not (the caller then packs in Python); -2^31 on a Python error.
 * need_writable != 0: the packer will remap feature3d IN PLACE (/root/reference/src/scale_calculator.py:390-394,:414), so
 * every feature3d buffer is requested WRITABLE — a read-only array (arr.flags.writeable = False, np.frombuffer over bytes, a
 * read-only memory map) makes the call return -(i+1), and the Python path the caller falls back to raises ValueError at the
 * assignment exactly as the reference does, instead of this library writing through a pointer it was not given for writing. */
long mvosr_py_frame_pointers(PyObject *f3s, PyObject *f2s, uint64_t *p3, uint64_t *p2, int32_t *n_points, int need_writable) {
    if (!PyList_Check(f3s) || !PyList_Check(f2s)) return -1;
    const Py_ssize_t n = PyList_GET_SIZE(f3s);
    if (PyList_GET_SIZE(f2s) != n) return -1;
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_buffer a, b;
        if (PyObject_GetBuffer(PyList_GET_ITEM(f3s, i), &a, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT | (need_writable ? PyBUF_WRITABLE : 0)) != 0) { PyErr_Clear(); return -(long)(i + 1); }
        if (PyObject_GetBuffer(PyList_GET_ITEM(f2s, i), &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); PyBuffer_Release(&a); return -(long)(i + 1); }
        const int ok = a.ndim == 2 && b.ndim == 2 && a.itemsize == 8 && b.itemsize == 8 && a.format && b.format &&
                       a.format[0] == 'd' && a.format[1] == 0 && b.format[0] == 'd' && b.format[1] == 0 &&
                       a.shape[1] == 3 && b.shape[1] == 2 && a.shape[0] == b.shape[0] && a.shape[0] < 0x7fffffff;
        if (ok) { p3[i] = (uint64_t)(uintptr_t)a.buf; p2[i] = (uint64_t)(uintptr_t)b.buf; n_points[i] = (int32_t)a.shape[0]; }
        PyBuffer_Release(&a);
        PyBuffer_Release(&b);
        if (!ok) return -(long)(i + 1);
    }
    return (long)n;
}
