import com.github.jonnylaw.model.CssmNative;

/** create -> init -> step x 3 -> filter through JNI, printed as hex doubles; tests/test_jvm_binding.py compares the lines with
 *  the same calls made through ctypes.  Model: BASELINE configs[0], Model.poisson(Sde.brownianMotion(1)),
 *  brownianParameter(0)(1)(0.01) stored as (m0, log c0, log sigma).  usage: CssmJniSmoke <particles> <seed> */
public final class CssmJniSmoke {
  public static void main(String[] a) {
    final long n = Long.parseLong(a[0]), seed = Long.parseLong(a[1]);
    // DescriptorBuilder's wire form (cssm_jni.c: unpack): [n_leaves, obs_kind, precision, df | sde, dim, f, period, harmonics,
    // has_scale, n_m0, n_c0, n_mu, n_phi, n_sigma], reals [scale, m0, c0, sigma]
    final int[] ints = {1, 0, 0, 0, /* leaf */ 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 1};
    final double[] reals = {0.0, 0.0, Math.log(1.0), Math.log(0.01)};
    final long h = CssmNative.create(ints, reals, n, seed, 0);
    final double[] t = {1.0, 2.0, 3.0, 4.5, 5.0}, y = {2.0, 0.0, 3.0, 1.0, 4.0};
    final byte[] has = {1, 1, 0, 1, 1};
    final double[] out = new double[2];
    CssmNative.init(h, 0.0);
    for (int s = 0; s < 3; ++s) {
      CssmNative.step(h, t[s], y[s], has[s] != 0, out);
      System.out.println("step " + Double.toHexString(out[0]) + " " + (int) out[1]);
    }
    final double[] cloud = new double[(int) n];
    CssmNative.particles(h, cloud);
    System.out.println("cloud " + Double.toHexString(cloud[0]) + " " + Double.toHexString(cloud[(int) n - 1]));
    final double[] path = new double[t.length + 1];
    System.out.println("filter " + Double.toHexString(CssmNative.filter(h, t, y, has, path)) + " " + Double.toHexString(path[t.length]));
    System.out.println("llFilter " + Double.toHexString(CssmNative.filter(h, t, y, has, null)));
    System.out.println("runKey " + Long.toUnsignedString(CssmNative.runKey(seed, 7)));
    try { CssmNative.create(new int[] {0, 0, 0, 0}, new double[] {}, n, seed, 0); System.out.println("error none"); }
    catch (RuntimeException e) { System.out.println("error " + e.getMessage()); }
    CssmNative.destroy(h);
  }
}
