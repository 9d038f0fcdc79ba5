/*
 * cssm_jni.c -- JNI glue between com.github.jonnylaw.model.CssmNative and libcssm_pf.
 *
 * The build image has no JDK (no jni.h); tests/test_jvm_binding.py probes for one on the GPU box and, where it finds
 * javac + jni.h, compiles this file with CssmNative.java and a small Java harness and compares what they return with the
 * ctypes path.  It is deliberately thin: every line of logic lives behind the C ABI of include/cssm_pf.h.
 *
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../../../include \
 *       cssm_jni.c -L<dir of libcssm_pf.so> -lcssm_pf -o libcssm_jni.so
 *
 * Error convention: a non-zero status becomes a RuntimeException carrying cssm_last_error(), which
 * matches the reference throwing from inside stepFilter (model/Sde.scala:214, model/Model.scala:150).
 */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "cssm_pf.h"

static void throw_last(JNIEnv* env) {
  jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
  (*env)->ThrowNew(env, ex, cssm_last_error());
}

/*
 * The descriptor arrives flattened from Scala (DescriptorBuilder in FilterGpu.scala):
 *   ints  : [n_leaves, obs_kind, lgcp_precision, obs_df, then per leaf: sde_kind, dim, f_kind, period, harmonics,
 *            has_scale, n_m0, n_c0, n_mu, n_phi, n_sigma]
 *   reals : per leaf: scale, m0[..], c0[..], mu[..], phi[..], sigma[..]   (STORED, unconstrained values)
 */
typedef struct { cssm_model_desc desc; cssm_leaf_desc* leaves; jdouble* reals; jint* ints; } owned_desc;

static int unpack(JNIEnv* env, jintArray ji, jdoubleArray jd, owned_desc* o) {
  o->ints = (*env)->GetIntArrayElements(env, ji, NULL);
  o->reals = (*env)->GetDoubleArrayElements(env, jd, NULL);
  const jint* a = o->ints;
  const double* r = o->reals;
  int n = a[0];
  o->leaves = (cssm_leaf_desc*)calloc((size_t)n, sizeof(cssm_leaf_desc));
  o->desc.n_leaves = n; o->desc.obs_kind = a[1]; o->desc.lgcp_precision = a[2]; o->desc.obs_df = a[3];
  o->desc.leaves = o->leaves;
  a += 4;
  for (int l = 0; l < n; ++l, a += 11) {
    cssm_leaf_desc* L = &o->leaves[l];
    L->sde_kind = a[0]; L->dim = a[1]; L->f_kind = a[2]; L->period = a[3]; L->harmonics = a[4]; L->has_scale = a[5];
    L->n_m0 = a[6]; L->n_c0 = a[7]; L->n_mu = a[8]; L->n_phi = a[9]; L->n_sigma = a[10];
    L->scale = *r++;
    L->m0 = r; r += L->n_m0;  L->c0 = r; r += L->n_c0;  L->mu = L->n_mu ? r : NULL; r += L->n_mu;
    L->phi = L->n_phi ? r : NULL; r += L->n_phi;  L->sigma = r; r += L->n_sigma;
  }
  return 0;
}

static void release(JNIEnv* env, jintArray ji, jdoubleArray jd, owned_desc* o) {
  free(o->leaves);
  (*env)->ReleaseIntArrayElements(env, ji, o->ints, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, jd, o->reals, JNI_ABORT);
}

JNIEXPORT jlong JNICALL Java_com_github_jonnylaw_model_CssmNative_create(JNIEnv* env, jclass c, jintArray ints,
                                                                         jdoubleArray reals, jlong n, jlong seed, jint device) {
  owned_desc o; cssm_pf* pf = NULL;
  unpack(env, ints, reals, &o);
  int rc = cssm_pf_create(&o.desc, (uint64_t)n, (uint64_t)seed, device, &pf);
  release(env, ints, reals, &o);
  if (rc) { throw_last(env); return 0; }
  return (jlong)(intptr_t)pf;
}

JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_destroy(JNIEnv* env, jclass c, jlong h) {
  cssm_pf_destroy((cssm_pf*)(intptr_t)h);
}

JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_setParams(JNIEnv* env, jclass c, jlong h, jintArray ints,
                                                                           jdoubleArray reals, jlong seed) {
  owned_desc o;
  unpack(env, ints, reals, &o);
  int rc = cssm_pf_set_params((cssm_pf*)(intptr_t)h, &o.desc);
  if (!rc) rc = cssm_pf_reseed((cssm_pf*)(intptr_t)h, (uint64_t)seed);
  release(env, ints, reals, &o);
  if (rc) throw_last(env);
}

JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_init(JNIEnv* env, jclass c, jlong h, jdouble t0) {
  if (cssm_pf_init((cssm_pf*)(intptr_t)h, t0)) throw_last(env);
}

JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_initFrom(JNIEnv* env, jclass c, jlong h, jdouble t0, jdoubleArray s) {
  jdouble* p = (*env)->GetDoubleArrayElements(env, s, NULL);
  int rc = cssm_pf_init_from((cssm_pf*)(intptr_t)h, t0, p);
  (*env)->ReleaseDoubleArrayElements(env, s, p, JNI_ABORT);
  if (rc) throw_last(env);
}

/* out[0] = ll, out[1] = ess */
JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_step(JNIEnv* env, jclass c, jlong h, jdouble t, jdouble y,
                                                                      jboolean hasObs, jdoubleArray out) {
  double ll; int32_t ess;
  if (cssm_pf_step((cssm_pf*)(intptr_t)h, t, y, hasObs ? 1 : 0, &ll, &ess)) { throw_last(env); return; }
  jdouble v[2] = {ll, (jdouble)ess};
  (*env)->SetDoubleArrayRegion(env, out, 0, 2, v);
}

/* llFilter / filter: path may be null (llFilter) */
JNIEXPORT jdouble JNICALL Java_com_github_jonnylaw_model_CssmNative_filter(JNIEnv* env, jclass c, jlong h, jdoubleArray jt,
                                                                           jdoubleArray jy, jbyteArray jhas, jdoubleArray jpath) {
  jsize T = (*env)->GetArrayLength(env, jt);
  jdouble* t = (*env)->GetDoubleArrayElements(env, jt, NULL);
  jdouble* y = (*env)->GetDoubleArrayElements(env, jy, NULL);
  jbyte* has = (*env)->GetByteArrayElements(env, jhas, NULL);
  jdouble* path = jpath ? (*env)->GetDoubleArrayElements(env, jpath, NULL) : NULL;
  double ll = 0.0;
  int rc = path ? cssm_pf_filter((cssm_pf*)(intptr_t)h, t, y, (const uint8_t*)has, (size_t)T, &ll, NULL, NULL, path)
                : cssm_pf_ll_filter((cssm_pf*)(intptr_t)h, t, y, (const uint8_t*)has, (size_t)T, &ll, NULL, NULL);
  (*env)->ReleaseDoubleArrayElements(env, jt, t, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, jy, y, JNI_ABORT);
  (*env)->ReleaseByteArrayElements(env, jhas, has, JNI_ABORT);
  if (path) (*env)->ReleaseDoubleArrayElements(env, jpath, path, 0);
  if (rc) throw_last(env);
  return ll;
}

/* PfState.particles on demand: SoA [d][N] */
JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_particles(JNIEnv* env, jclass c, jlong h, jdoubleArray out) {
  jdouble* p = (*env)->GetDoubleArrayElements(env, out, NULL);
  int rc = cssm_pf_get_particles((cssm_pf*)(intptr_t)h, p);
  (*env)->ReleaseDoubleArrayElements(env, out, p, 0);
  if (rc) throw_last(env);
}

/* Resample[A] shim for the other resamplers (kind = CSSM_RESAMPLE_*; uniforms from the Philox streams of (seed, step)) */
JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_resample(JNIEnv* env, jclass c, jint kind, jdoubleArray jw, jdouble u,
                                                                          jlong seed, jint step, jintArray janc, jint device) {
  jsize n = (*env)->GetArrayLength(env, jw);
  jdouble* w = (*env)->GetDoubleArrayElements(env, jw, NULL);
  jint* a = (*env)->GetIntArrayElements(env, janc, NULL);
  int rc = cssm_resample(kind, w, (size_t)n, u, (uint64_t)seed, (uint32_t)step, (uint32_t*)a, device);
  (*env)->ReleaseDoubleArrayElements(env, jw, w, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, janc, a, 0);
  if (rc) throw_last(env);
}

/* Resample[A] shim: ancestor indices for host-side weights */
JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_resampleSystematic(JNIEnv* env, jclass c, jdoubleArray jw,
                                                                                    jdouble u, jintArray janc, jint device) {
  jsize n = (*env)->GetArrayLength(env, jw);
  jdouble* w = (*env)->GetDoubleArrayElements(env, jw, NULL);
  jint* a = (*env)->GetIntArrayElements(env, janc, NULL);
  int rc = cssm_resample_systematic(w, (size_t)n, u, (uint32_t*)a, device);
  (*env)->ReleaseDoubleArrayElements(env, jw, w, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, janc, a, 0);
  if (rc) throw_last(env);
}

/* The Philox key of the run-th filter run under one user seed (a PRF of (seed, run); include/cssm_numerics.h) */
JNIEXPORT jlong JNICALL Java_com_github_jonnylaw_model_CssmNative_runKey(JNIEnv* env, jclass c, jlong seed, jlong run) {
  return (jlong)cssm_pf_run_key((uint64_t)seed, (uint64_t)run);
}

/* Filter(mod, resample): the resampler of the device path (CSSM_OPT_RESAMPLER) */
JNIEXPORT void JNICALL Java_com_github_jonnylaw_model_CssmNative_setResampler(JNIEnv* env, jclass c, jlong h, jint kind) {
  if (cssm_pf_set_option((cssm_pf*)(intptr_t)h, CSSM_OPT_RESAMPLER, kind)) throw_last(env);
}
