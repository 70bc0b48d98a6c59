/* _asr_fastcall: a minimal CPython extension that calls the C ABI of libasr_hip.so (include/asr_hip.h).
 *
 * Why: the training step issues 350 (CTC config) to 600 (joint config) launches, and a ctypes foreign call with
 * 15-25 arguments costs ~6.4 us of host time each - the joint step was host-bound (5.4 ms of enqueue for 5.5 ms of
 * GPU work).  A vectorcall into this module converts the same arguments in ~0.3 us.  It is host plumbing exactly
 * like ctypes: it knows nothing about the kernels, takes the function ADDRESS (from ctypes / dlsym) and a signature
 * string, and every argument is a plain pointer / integer / float of the C ABI.
 *
 * Calling convention (x86-64 System V, the only host this image has): integer-class arguments (pointers, int,
 * uint32_t, size_t) and float arguments are assigned to their register files independently, floats never reach the
 * stack while there are at most 8 of them, and stack slots are 8 bytes each in argument order.  A call through
 *     long fn(long i0 .. i27, float f0 .. f7)
 * with the integer-class arguments in their original order and the floats in theirs therefore places every argument
 * where the real prototype expects it, whatever the interleaving; surplus trailing arguments are ignored by the
 * callee.  make() refuses signatures outside these limits.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if !defined(__x86_64__) || defined(_WIN32)
#error "fastcall.c relies on the x86-64 System V calling convention"
#endif

#define MAX_INT 28
#define MAX_FLT 8
#define MAX_ARGS 36

typedef long (*generic_fn)(long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long,
                           long, long, long, long, long, long, long, float, float, float, float, float, float, float, float);

typedef struct {
    PyObject_HEAD
    vectorcallfunc vectorcall;
    generic_fn fn;
    int nargs;
    int ret_size_t;          /* 'Z': size_t result, otherwise int */
    char sig[MAX_ARGS + 1];  /* one of P I U Z F per argument */
    char name[64];
} FastFn;

static int as_intclass(PyObject* o, char kind, long* out, const char* fname, int idx) {
    if (o == Py_None) {
        if (kind != 'P') {
            PyErr_Format(PyExc_TypeError, "%s: argument %d: None is only valid for a pointer", fname, idx);
            return -1;
        }
        *out = 0;
        return 0;
    }
    if (PyLong_CheckExact(o) || PyLong_Check(o)) {
        if (kind == 'P' || kind == 'Z') {
            unsigned long long v = PyLong_AsUnsignedLongLong(o);
            if (v == (unsigned long long)-1 && PyErr_Occurred()) return -1;
            *out = (long)v;
        } else {
            long v = PyLong_AsLong(o);
            if (v == -1 && PyErr_Occurred()) return -1;
            if (kind == 'I' && (v > INT32_MAX || v < INT32_MIN)) {
                PyErr_Format(PyExc_OverflowError, "%s: argument %d does not fit an int", fname, idx);
                return -1;
            }
            if (kind == 'U' && (v < 0 || v > (long)UINT32_MAX)) {
                PyErr_Format(PyExc_OverflowError, "%s: argument %d does not fit a uint32_t", fname, idx);
                return -1;
            }
            *out = v;
        }
        return 0;
    }
    if (kind == 'I' && PyBool_Check(o)) {
        *out = o == Py_True;
        return 0;
    }
    if (kind == 'P') { /* ctypes.c_void_p / c_char_p instances and the like: their .value */
        PyObject* v = PyObject_GetAttrString(o, "value");
        if (v) {
            int rc = as_intclass(v, 'P', out, fname, idx);
            Py_DECREF(v);
            return rc;
        }
        PyErr_Clear();
    }
    PyErr_Format(PyExc_TypeError, "%s: argument %d: expected an int%s, got %.80s", fname, idx, kind == 'P' ? " address or None" : "", Py_TYPE(o)->tp_name);
    return -1;
}

static PyObject* fastfn_vectorcall(PyObject* self_, PyObject* const* args, size_t nargsf, PyObject* kwnames) {
    FastFn* self = (FastFn*)self_;
    const Py_ssize_t n = PyVectorcall_NARGS(nargsf);
    if (kwnames && PyTuple_GET_SIZE(kwnames)) {
        PyErr_Format(PyExc_TypeError, "%s takes no keyword arguments", self->name);
        return NULL;
    }
    if (n != self->nargs) {
        PyErr_Format(PyExc_TypeError, "%s takes %d arguments (%zd given)", self->name, self->nargs, n);
        return NULL;
    }
    long iv[MAX_INT] = {0};
    float fv[MAX_FLT] = {0};
    int ni = 0, nf = 0;
    for (Py_ssize_t a = 0; a < n; ++a) {
        const char kind = self->sig[a];
        if (kind == 'F') {
            double d = PyFloat_AsDouble(args[a]);
            if (d == -1.0 && PyErr_Occurred()) return NULL;
            fv[nf++] = (float)d;
        } else {
            if (as_intclass(args[a], kind, &iv[ni++], self->name, (int)a) < 0) return NULL;
        }
    }
    long r;
    Py_BEGIN_ALLOW_THREADS
    r = self->fn(iv[0], iv[1], iv[2], iv[3], iv[4], iv[5], iv[6], iv[7], iv[8], iv[9], iv[10], iv[11], iv[12], iv[13], iv[14], iv[15], iv[16], iv[17],
                 iv[18], iv[19], iv[20], iv[21], iv[22], iv[23], iv[24], iv[25], iv[26], iv[27], fv[0], fv[1], fv[2], fv[3], fv[4], fv[5], fv[6], fv[7]);
    Py_END_ALLOW_THREADS
    if (self->ret_size_t) return PyLong_FromUnsignedLongLong((unsigned long long)r);
    return PyLong_FromLong((long)(int)r);
}

static PyObject* fastfn_repr(PyObject* self_) {
    FastFn* self = (FastFn*)self_;
    return PyUnicode_FromFormat("<fastcall %s(%s) at %p>", self->name, self->sig, (void*)self->fn);
}

static PyTypeObject FastFnType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "_asr_fastcall.FastFn",
    .tp_basicsize = sizeof(FastFn),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL,
    .tp_vectorcall_offset = offsetof(FastFn, vectorcall),
    .tp_call = PyVectorcall_Call,
    .tp_repr = fastfn_repr,
    .tp_doc = "callable bound to one C-ABI entry point",
};

/* make(address, name, signature, restype) -> FastFn.  signature: string over P (pointer) I (int) U (uint32_t)
 * Z (size_t) F (float); restype 'I' or 'Z'. */
static PyObject* make(PyObject* mod, PyObject* args) {
    unsigned long long addr;
    const char *name, *sig, *res;
    if (!PyArg_ParseTuple(args, "Ksss", &addr, &name, &sig, &res)) return NULL;
    const size_t n = strlen(sig);
    int ni = 0, nf = 0;
    if (!addr) {
        PyErr_Format(PyExc_ValueError, "%s: null function address", name);
        return NULL;
    }
    if (n > MAX_ARGS) {
        PyErr_Format(PyExc_ValueError, "%s: too many arguments", name);
        return NULL;
    }
    for (size_t i = 0; i < n; ++i) {
        if (sig[i] == 'F') ++nf;
        else if (strchr("PIUZ", sig[i])) ++ni;
        else {
            PyErr_Format(PyExc_ValueError, "%s: unknown argument kind '%c'", name, sig[i]);
            return NULL;
        }
    }
    if (ni > MAX_INT || nf > MAX_FLT || (res[0] != 'I' && res[0] != 'Z')) {
        PyErr_Format(PyExc_ValueError, "%s: signature outside what the trampoline covers (%d integer-class, %d float arguments)", name, ni, nf);
        return NULL;
    }
    FastFn* f = PyObject_New(FastFn, &FastFnType);
    if (!f) return NULL;
    f->vectorcall = fastfn_vectorcall;
    f->fn = (generic_fn)(uintptr_t)addr;
    f->nargs = (int)n;
    f->ret_size_t = res[0] == 'Z';
    strcpy(f->sig, sig);
    strncpy(f->name, name, sizeof(f->name) - 1);
    f->name[sizeof(f->name) - 1] = 0;
    return (PyObject*)f;
}

static PyMethodDef methods[] = {{"make", make, METH_VARARGS, "make(address, name, signature, restype) -> callable"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_asr_fastcall", "fast foreign calls into libasr_hip.so", -1, methods};

PyMODINIT_FUNC PyInit__asr_fastcall(void) {
    if (PyType_Ready(&FastFnType) < 0) return NULL;
    PyObject* m = PyModule_Create(&moddef);
    if (!m) return NULL;
    Py_INCREF(&FastFnType);
    PyModule_AddObject(m, "FastFn", (PyObject*)&FastFnType);
    return m;
}
