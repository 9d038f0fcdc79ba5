package com.github.jonnylaw.model;

/**
 * JNI entry points of libcssm_jni.so (jvm/src/main/c/cssm_jni.c) over include/cssm_pf.h.
 *
 * A Java class with STATIC natives on purpose: the glue's symbols are Java_com_github_jonnylaw_model_CssmNative_<method>
 * taking a jclass.  A Scala `object CssmNative { @native def ... }` would put the natives on the module class
 * `CssmNative$` (JNI name ..._CssmNative_00024_<method>, instance methods) and fail to link at the first call; sbt compiles
 * mixed Java / Scala sources, and Scala calls these as `CssmNative.step(...)` unchanged (FilterGpu.scala).
 * tests/test_jvm_binding.py compiles this file, the glue and jvm/src/test/java/CssmJniSmoke.java wherever a JDK exists.
 */
public final class CssmNative {
  static { System.loadLibrary("cssm_jni"); }
  private CssmNative() {}

  public static native long create(int[] ints, double[] reals, long n, long seed, int device);
  public static native void destroy(long handle);
  public static native void setParams(long handle, int[] ints, double[] reals, long seed);
  public static native void init(long handle, double t0);
  public static native void initFrom(long handle, double t0, double[] state);
  /** out[0] = ll, out[1] = ess */
  public static native void step(long handle, double t, double y, boolean hasObs, double[] out);
  /** llFilter (path == null) / filter (path: (T + 1) * d doubles) */
  public static native double filter(long handle, double[] t, double[] y, byte[] has, double[] path);
  public static native void particles(long handle, double[] out);
  public static native void resampleSystematic(double[] w, double u, int[] anc, int device);
  public static native void resample(int kind, double[] w, double u, long seed, int step, int[] anc, int device);
  /** cssm_pf_run_key: the Philox key of the run-th filter run under one user seed -- a PRF of (seed, run), never seed + run. */
  public static native long runKey(long seed, long run);
  /** cssm_pf_set_option(handle, CSSM_OPT_RESAMPLER, kind). */
  public static native void setResampler(long handle, int kind);
}
