/* host/bind.c — CPython module `audiosync` for the MI355X build.
 *
 * Same eight functions, names, arities and return types as the reference module
 * (src/bind.c:22-75: run, pause, resume, abort, status, setup, get_debug, set_debug; `run`
 * returns (lag:int, success:bool) like the "lO" of src/bind.c:107), every C call made with the
 * GIL released like src/bind.c:102-104.  Two additions, because the reference never exposed
 * the hot path to Python:
 *   cross_correlation(source, sample) -> (ret, lag, coefficient)   float64 buffers, len(source) == 2*len(sample)
 *   set_feed(source, sample, frames_per_ms=0)                      tracks that run() will consume
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <stdlib.h>
#include <string.h>

#include <audiosync/audiosync.h>
#include <audiosync/cross_correlation.h>

/* copies of the fed tracks: run() may outlive the Python objects that were passed in */
static double *feed_source_copy, *feed_sample_copy;

static PyObject *mod_run(PyObject *self, PyObject *args)
{
    UNUSED(self);
    const char *title;
    if (!PyArg_ParseTuple(args, "s", &title)) return NULL;
    long lag = 0;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = audiosync_run(title, &lag);
    Py_END_ALLOW_THREADS
    return Py_BuildValue("lO", lag, rc == 0 ? Py_True : Py_False);
}

static PyObject *mod_pause(PyObject *self, PyObject *noargs)
{
    UNUSED(self); UNUSED(noargs);
    Py_BEGIN_ALLOW_THREADS
    audiosync_pause();
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

static PyObject *mod_resume(PyObject *self, PyObject *noargs)
{
    UNUSED(self); UNUSED(noargs);
    Py_BEGIN_ALLOW_THREADS
    audiosync_resume();
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

static PyObject *mod_abort(PyObject *self, PyObject *noargs)
{
    UNUSED(self); UNUSED(noargs);
    Py_BEGIN_ALLOW_THREADS
    audiosync_abort();
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

static PyObject *mod_status(PyObject *self, PyObject *noargs)
{
    UNUSED(self); UNUSED(noargs);
    global_status_t st;
    Py_BEGIN_ALLOW_THREADS
    st = audiosync_status();
    Py_END_ALLOW_THREADS
    return PyUnicode_FromString(status_to_string(st));
}

static PyObject *mod_setup(PyObject *self, PyObject *args)
{
    UNUSED(self);
    const char *stream_name;
    if (!PyArg_ParseTuple(args, "s", &stream_name)) return NULL;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = audiosync_setup(stream_name);
    Py_END_ALLOW_THREADS
    return PyBool_FromLong(rc == 0);
}

static PyObject *mod_get_debug(PyObject *self, PyObject *noargs)
{
    UNUSED(self); UNUSED(noargs);
    return PyBool_FromLong(audiosync_get_debug());
}

static PyObject *mod_set_debug(PyObject *self, PyObject *args)
{
    UNUSED(self);
    int flag;
    if (!PyArg_ParseTuple(args, "p", &flag)) return NULL;
    audiosync_set_debug(flag);
    Py_RETURN_NONE;
}

static int as_doubles(PyObject *obj, Py_buffer *view, const char *what)
{
    if (PyObject_GetBuffer(obj, view, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) != 0) return -1;
    if (view->itemsize != (Py_ssize_t)sizeof(double) || view->format == NULL || strcmp(view->format, "d") != 0) {
        PyBuffer_Release(view);
        PyErr_Format(PyExc_TypeError, "%s must be a contiguous buffer of float64", what);
        return -1;
    }
    return 0;
}

static PyObject *mod_cross_correlation(PyObject *self, PyObject *args)
{
    UNUSED(self);
    PyObject *osrc, *osmp;
    if (!PyArg_ParseTuple(args, "OO", &osrc, &osmp)) return NULL;
    Py_buffer src, smp;
    if (as_doubles(osrc, &src, "source") != 0) return NULL;
    if (as_doubles(osmp, &smp, "sample") != 0) { PyBuffer_Release(&src); return NULL; }
    const Py_ssize_t n = smp.len / (Py_ssize_t)sizeof(double);
    if (n < 1 || src.len != 2 * smp.len) {
        PyBuffer_Release(&src); PyBuffer_Release(&smp);
        PyErr_SetString(PyExc_ValueError, "len(source) must be 2 * len(sample) and sample must not be empty");
        return NULL;
    }
    long lag = 0;
    double coef = 0.0;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = cross_correlation((double *)src.buf, (double *)smp.buf, (size_t)n, &lag, &coef);
    Py_END_ALLOW_THREADS
    PyBuffer_Release(&src); PyBuffer_Release(&smp);
    return Py_BuildValue("ild", rc, lag, coef);
}

static PyObject *mod_set_feed(PyObject *self, PyObject *args)
{
    UNUSED(self);
    PyObject *osrc, *osmp;
    unsigned int frames_per_ms = 0;
    if (!PyArg_ParseTuple(args, "OO|I", &osrc, &osmp, &frames_per_ms)) return NULL;
    if (audiosync_status() != IDLE_ST) {
        PyErr_SetString(PyExc_RuntimeError, "set_feed() while a run is in progress");
        return NULL;
    }
    Py_buffer src, smp;
    if (as_doubles(osrc, &src, "source") != 0) return NULL;
    if (as_doubles(osmp, &smp, "sample") != 0) { PyBuffer_Release(&src); return NULL; }
    double *s = malloc(src.len ? (size_t)src.len : 1), *t = malloc(smp.len ? (size_t)smp.len : 1);
    if (!s || !t) {
        free(s); free(t);
        PyBuffer_Release(&src); PyBuffer_Release(&smp);
        return PyErr_NoMemory();
    }
    memcpy(s, src.buf, (size_t)src.len);
    memcpy(t, smp.buf, (size_t)smp.len);
    audiosync_set_feed(s, (size_t)src.len / sizeof(double), t, (size_t)smp.len / sizeof(double), frames_per_ms);
    free(feed_source_copy); free(feed_sample_copy);
    feed_source_copy = s; feed_sample_copy = t;
    PyBuffer_Release(&src); PyBuffer_Release(&smp);
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    { "run", mod_run, METH_VARARGS, "run(title) -> (lag_ms, success). One run at a time." },
    { "pause", mod_pause, METH_NOARGS, "Pause the current run. Thread-safe." },
    { "resume", mod_resume, METH_NOARGS, "Resume a paused run. Thread-safe." },
    { "abort", mod_abort, METH_NOARGS, "Abort the current run. Thread-safe." },
    { "status", mod_status, METH_NOARGS, "'idle', 'running', 'paused' or 'aborting'. Thread-safe." },
    { "setup", mod_setup, METH_VARARGS, "setup(stream_name) -> bool. Always False here (no PulseAudio)." },
    { "get_debug", mod_get_debug, METH_NOARGS, "Debug logging on? Thread-safe." },
    { "set_debug", mod_set_debug, METH_VARARGS, "set_debug(flag). Thread-safe." },
    { "cross_correlation", mod_cross_correlation, METH_VARARGS,
      "cross_correlation(source, sample) -> (ret, lag, coefficient) on the GPU; float64 buffers." },
    { "set_feed", mod_set_feed, METH_VARARGS,
      "set_feed(source, sample, frames_per_ms=0): the tracks run() will 'download' and 'record'." },
    { NULL, NULL, 0, NULL }
};

static struct PyModuleDef moduledef = {
    PyModuleDef_HEAD_INIT, "audiosync",
    "Audio synchronization (FFT cross-correlation on MI355X) with the vidify-audiosync interface.",
    -1, methods, NULL, NULL, NULL, NULL
};

PyMODINIT_FUNC PyInit_audiosync(void) { return PyModule_Create(&moduledef); }
