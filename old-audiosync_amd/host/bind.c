/* host/bind.c — CPython module `audiosync` for the MI355X build.
 *
 * Same eight functions, names, arities and return types as the reference module
 * (src/bind.c:22-75: run, pause, resume, abort, status, setup, get_debug, set_debug; `run`
 * returns (lag:int, success:bool) like the "lO" of src/bind.c:107), every C call made with the
 * GIL released like src/bind.c:102-104.  Two additions, because the reference never exposed
 * the hot path to Python:
 *   cross_correlation(source, sample) -> (ret, lag, coefficient)   float64 or float32 buffers,
 *                                                                   len(source) == 2*len(sample)
 *   cross_correlation_batch(sources, samples) -> (rets, lags, coefficients)
 *                                                                   B pairs: sources [B][2N], samples [B][N]
 *                                                                   (any C-contiguous buffers of that many
 *                                                                   float32 or float64 values; 1-D or 2-D)
 *   set_feed(source, sample, frames_per_ms=0)                      tracks that run() will consume
 *   set_feed_files(source_path, sample_path)                       the same from f64le files / FIFOs
 * float64 goes through cross_correlation(double*) (transforms in float32, Pearson on the doubles);
 * float32 goes through asx_xcorr_batch_f32 (one launch group per call, the batched path of the benchmark).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <audiosync/audiosync.h>
#include <audiosync/cross_correlation.h>
#include <audiosync/xcorr_hip.h>

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

/* float64 ("d") or float32 ("f") C-contiguous buffer; returns the item size (8 / 4) or -1 */
static int as_reals(PyObject *obj, Py_buffer *view, const char *what)
{
    if (PyObject_GetBuffer(obj, view, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) != 0) return -1;
    const char *f = view->format ? view->format : "";
    if (*f == '=' || *f == '<' || *f == '@') f++;
    if (view->itemsize == (Py_ssize_t)sizeof(double) && strcmp(f, "d") == 0) return 8;
    if (view->itemsize == (Py_ssize_t)sizeof(float) && strcmp(f, "f") == 0) return 4;
    PyBuffer_Release(view);
    PyErr_Format(PyExc_TypeError, "%s must be a contiguous buffer of float64 or float32", what);
    return -1;
}

/* the float32 entry points run on a plan kept between calls (sample_len and batch capacity) */
static pthread_mutex_t f32_mutex = PTHREAD_MUTEX_INITIALIZER;
static asx_plan *f32_plan;
static size_t f32_plan_n, f32_plan_cap;
static int f32_plan_dev = -1;

/* with the GIL released */
static int run_f32(const float *src, const float *smp, size_t n, size_t batch, int64_t *lag, double *coef, int32_t *ret)
{
    int rc = -1;
    pthread_mutex_lock(&f32_mutex);
    /* keyed on the sample length, the batch capacity AND the current HIP device (like the float64 path's plan cache):
     * after torch.cuda.set_device() the call must not keep running on the old device.  The capacity grows
     * geometrically, so a caller whose batches creep up does not rebuild the plan at every call. */
    const int dev = asx_current_device();
    if (!f32_plan || f32_plan_n != n || f32_plan_cap < batch || f32_plan_dev != dev) {
        size_t cap = batch;
        if (f32_plan && f32_plan_n == n && f32_plan_dev == dev && cap < 2 * f32_plan_cap) cap = 2 * f32_plan_cap;
        if (f32_plan) asx_plan_destroy(f32_plan);
        f32_plan = asx_plan_create(n, cap, dev);
        if (!f32_plan && cap != batch) { cap = batch; f32_plan = asx_plan_create(n, cap, dev); }
        f32_plan_n = n;
        f32_plan_cap = f32_plan ? cap : 0;
        f32_plan_dev = dev;
    }
    if (f32_plan) rc = asx_xcorr_batch_f32(f32_plan, src, smp, batch, lag, coef, ret);
    pthread_mutex_unlock(&f32_mutex);
    return rc;
}

static PyObject *mod_cross_correlation(PyObject *self, PyObject *args)
{
    UNUSED(self);
    PyObject *osrc, *osmp;
    if (!PyArg_ParseTuple(args, "OO", &osrc, &osmp)) return NULL;
    Py_buffer src, smp;
    const int ws = as_reals(osrc, &src, "source");
    if (ws < 0) return NULL;
    const int wt = as_reals(osmp, &smp, "sample");
    if (wt < 0) { PyBuffer_Release(&src); return NULL; }
    if (ws != wt) {
        PyBuffer_Release(&src); PyBuffer_Release(&smp);
        PyErr_SetString(PyExc_TypeError, "source and sample must have the same element type");
        return NULL;
    }
    const Py_ssize_t n = smp.len / wt;
    if (n < 1 || src.len != 2 * smp.len) {
        PyBuffer_Release(&src); PyBuffer_Release(&smp);
        PyErr_SetString(PyExc_ValueError, "len(source) must be 2 * len(sample) and sample must not be empty");
        return NULL;
    }
    long lag = 0;
    double coef = 0.0;
    int rc;
    if (ws == 8) {
        Py_BEGIN_ALLOW_THREADS
        rc = cross_correlation((double *)src.buf, (double *)smp.buf, (size_t)n, &lag, &coef);
        Py_END_ALLOW_THREADS
    } else {
        int64_t l64 = 0;
        int32_t r32 = -1;
        int call;
        Py_BEGIN_ALLOW_THREADS
        call = run_f32((const float *)src.buf, (const float *)smp.buf, (size_t)n, 1, &l64, &coef, &r32);
        Py_END_ALLOW_THREADS
        if (call != 0) {
            PyBuffer_Release(&src); PyBuffer_Release(&smp);
            PyErr_Format(PyExc_RuntimeError, "audiosync: %s", asx_last_error());
            return NULL;
        }
        rc = r32;
        lag = (long)l64;
    }
    PyBuffer_Release(&src); PyBuffer_Release(&smp);
    return Py_BuildValue("ild", rc, lag, coef);
}

static PyObject *mod_cross_correlation_batch(PyObject *self, PyObject *args)
{
    UNUSED(self);
    PyObject *osrc, *osmp;
    Py_ssize_t batch_arg = -1;
    if (!PyArg_ParseTuple(args, "OO|n", &osrc, &osmp, &batch_arg)) return NULL;
    Py_buffer src, smp;
    const int ws = as_reals(osrc, &src, "sources");
    if (ws < 0) return NULL;
    const int wt = as_reals(osmp, &smp, "samples");
    if (wt < 0) { PyBuffer_Release(&src); return NULL; }
    /* batch: leading dimension of a 2-D samples buffer, or the third argument for flat buffers */
    Py_ssize_t batch = batch_arg;
    if (batch < 0) batch = (smp.ndim >= 2 && smp.shape) ? smp.shape[0] : 1;
    const Py_ssize_t total = smp.len / wt;
    PyObject *result = NULL;
    int64_t *lag = NULL;
    double *coef = NULL;
    int32_t *ret = NULL;
    if (ws != wt) {
        PyErr_SetString(PyExc_TypeError, "sources and samples must have the same element type");
        goto done;
    }
    if (batch < 1 || total < batch || total % batch != 0 || src.len != 2 * smp.len) {
        PyErr_SetString(PyExc_ValueError, "samples must hold batch * N values and sources batch * 2N");
        goto done;
    }
    const size_t n = (size_t)(total / batch);
    lag = malloc(sizeof(int64_t) * (size_t)batch);
    coef = malloc(sizeof(double) * (size_t)batch);
    ret = malloc(sizeof(int32_t) * (size_t)batch);
    if (!lag || !coef || !ret) { PyErr_NoMemory(); goto done; }
    int call = 0;
    Py_BEGIN_ALLOW_THREADS
    if (ws == 4) {
        call = run_f32((const float *)src.buf, (const float *)smp.buf, n, (size_t)batch, lag, coef, ret);
    } else {
        for (Py_ssize_t b = 0; b < batch; b++) {
            long l = 0;
            coef[b] = 0.0;
            ret[b] = cross_correlation((double *)src.buf + (size_t)b * 2 * n, (double *)smp.buf + (size_t)b * n, n, &l, &coef[b]);
            lag[b] = l;
        }
    }
    Py_END_ALLOW_THREADS
    if (call != 0) {
        PyErr_Format(PyExc_RuntimeError, "audiosync: %s", asx_last_error());
        goto done;
    }
    PyObject *rets = PyList_New(batch), *lags = PyList_New(batch), *coefs = PyList_New(batch);
    if (rets && lags && coefs) {
        int ok = 1;
        for (Py_ssize_t b = 0; b < batch && ok; b++) {
            PyObject *r = PyLong_FromLong(ret[b]), *l = PyLong_FromLongLong((long long)lag[b]), *c = PyFloat_FromDouble(coef[b]);
            if (!r || !l || !c) {
                Py_XDECREF(r); Py_XDECREF(l); Py_XDECREF(c);
                ok = 0;
                break;
            }
            PyList_SET_ITEM(rets, b, r);
            PyList_SET_ITEM(lags, b, l);
            PyList_SET_ITEM(coefs, b, c);
        }
        if (ok) result = PyTuple_Pack(3, rets, lags, coefs);
    }
    Py_XDECREF(rets); Py_XDECREF(lags); Py_XDECREF(coefs);
done:
    free(lag); free(coef); free(ret);
    PyBuffer_Release(&src); PyBuffer_Release(&smp);
    return result;
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
    if (audiosync_set_feed(s, (size_t)src.len / sizeof(double), t, (size_t)smp.len / sizeof(double), frames_per_ms) != 0) {
        /* a run started between the status check above and here: the library refused */
        free(s); free(t);
        PyBuffer_Release(&src); PyBuffer_Release(&smp);
        PyErr_SetString(PyExc_RuntimeError, "set_feed() while a run is in progress");
        return NULL;
    }
    free(feed_source_copy); free(feed_sample_copy);
    feed_source_copy = s; feed_sample_copy = t;
    PyBuffer_Release(&src); PyBuffer_Release(&smp);
    Py_RETURN_NONE;
}

static PyObject *mod_set_feed_files(PyObject *self, PyObject *args)
{
    UNUSED(self);
    const char *src_path, *smp_path;
    if (!PyArg_ParseTuple(args, "ss", &src_path, &smp_path)) return NULL;
    if (audiosync_status() != IDLE_ST) {
        PyErr_SetString(PyExc_RuntimeError, "set_feed_files() while a run is in progress");
        return NULL;
    }
    if (audiosync_set_feed_files(src_path, smp_path) != 0) {
        PyErr_SetString(PyExc_RuntimeError, "set_feed_files() failed");
        return NULL;
    }
    free(feed_source_copy); free(feed_sample_copy);
    feed_source_copy = NULL; feed_sample_copy = NULL;
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
      "cross_correlation(source, sample) -> (ret, lag, coefficient) on the GPU; float64 or float32 buffers." },
    { "cross_correlation_batch", mod_cross_correlation_batch, METH_VARARGS,
      "cross_correlation_batch(sources, samples[, batch]) -> (rets, lags, coefficients); sources [B][2N], "
      "samples [B][N], float32 (one batched launch) or float64 (the double ABI per pair)." },
    { "set_feed", mod_set_feed, METH_VARARGS,
      "set_feed(source, sample, frames_per_ms=0): the tracks run() will 'download' and 'record'." },
    { "set_feed_files", mod_set_feed_files, METH_VARARGS,
      "set_feed_files(source_path, sample_path): files or FIFOs of f64le mono frames at 48 kHz (what the "
      "reference reads from `ffmpeg -f f64le`) that run() will consume." },
    { NULL, NULL, 0, NULL }
};

static struct PyModuleDef moduledef = {
    PyModuleDef_HEAD_INIT, "audiosync",
    "Audio synchronization (FFT cross-correlation on MI355X) with the vidify-audiosync interface.",
    -1, methods, NULL, NULL, NULL, NULL
};

PyMODINIT_FUNC PyInit_audiosync(void) { return PyModule_Create(&moduledef); }
