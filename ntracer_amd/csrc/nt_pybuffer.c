/* nt_pybuffer.c -- the buffer protocol for the Python mirror's Vector and Color (the reference's types expose their
 * floats through it: obj_Vector / obj_Color tp_as_buffer, src/ntracer_body.hpp, src/render.cpp; its own test is
 * lib/ntracer/tests/test.py:294-300 -- list(memoryview(v)) == list(v)).  Pure Python cannot provide tp_as_buffer before
 * 3.12, so this is a one-type CPython module: `FloatBuffer`, a base class whose bf_getbuffer asks the instance for an
 * object that holds the floats (`_float_buffer()`, a read-only float32 array) and hands out that object's buffer.
 * Host-side convenience only; nothing on the render path uses it. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static int floatbuffer_getbuffer(PyObject *self, Py_buffer *view, int flags) {
    PyObject *inner = PyObject_CallMethod(self, "_float_buffer", NULL);
    if (!inner) return -1;
    int r = PyObject_GetBuffer(inner, view, flags);      /* view->obj = inner (a new reference held by the view) */
    Py_DECREF(inner);
    return r;
}

static PyBufferProcs floatbuffer_as_buffer = {floatbuffer_getbuffer, NULL};

static PyTypeObject FloatBufferType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "ntracer_amd._pybuffer.FloatBuffer",
    .tp_basicsize = sizeof(PyObject),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE,
    .tp_doc = "base class giving Vector / Color the buffer protocol (floats, read-only)",
    .tp_as_buffer = &floatbuffer_as_buffer,
    .tp_new = PyType_GenericNew,
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_pybuffer", "buffer protocol for Vector / Color", -1, NULL, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pybuffer(void) {
    if (PyType_Ready(&FloatBufferType) < 0) return NULL;
    PyObject *m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    Py_INCREF(&FloatBufferType);
    if (PyModule_AddObject(m, "FloatBuffer", (PyObject *)&FloatBufferType) < 0) {
        Py_DECREF(&FloatBufferType);
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
