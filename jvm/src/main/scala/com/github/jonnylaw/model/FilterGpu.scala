package com.github.jonnylaw.model

// SOURCE ONLY (no JDK/sbt in the build image): the Scala side of the drop-in boundary.
// It lives in package com.github.jonnylaw.model because the model case classes it pattern-matches
// (PoissonModel, SeasonalModel, LinearModel, LogGaussianCox, BrownianMotion, GenBrownianMotion,
// OuProcess) are `private final` there (Model.scala:144-363, Sde.scala:69,98,129).

import akka.NotUsed
import akka.stream.scaladsl.Flow
import breeze.linalg.DenseVector
import cats.data.Reader

/** JNI entry points of libcssm_jni.so (jvm/src/main/c/cssm_jni.c) over include/cssm_pf.h. */
object CssmNative {
  System.loadLibrary("cssm_jni")
  @native def create(ints: Array[Int], reals: Array[Double], n: Long, seed: Long, device: Int): Long
  @native def destroy(handle: Long): Unit
  @native def setParams(handle: Long, ints: Array[Int], reals: Array[Double], seed: Long): Unit
  @native def init(handle: Long, t0: Double): Unit
  @native def initFrom(handle: Long, t0: Double, state: Array[Double]): Unit
  @native def step(handle: Long, t: Double, y: Double, hasObs: Boolean, out: Array[Double]): Unit
  @native def filter(handle: Long, t: Array[Double], y: Array[Double], has: Array[Byte], path: Array[Double]): Double
  @native def particles(handle: Long, out: Array[Double]): Unit
  @native def resampleSystematic(w: Array[Double], u: Double, anc: Array[Int], device: Int): Unit
  @native def resample(kind: Int, w: Array[Double], u: Double, seed: Long, step: Int, anc: Array[Int], device: Int): Unit
}

/** Flattens a parameterised composed model into the cssm_model_desc wire form (see cssm_jni.c). */
object DescriptorBuilder {
  private val (brownian, genBrownian, ou) = (0, 1, 2)
  private val (fFirst, fSeasonal) = (0, 1)
  private val (poisson, gaussian, lgcp) = (0, 1, 2)

  /** (unparameterised leaf kinds, parameters) => (ints, reals).  `leaves` lists, left to right,
    * (obs/f kind, sde kind, dimension, period, harmonics) exactly as the user composed them with |+|. */
  case class LeafSpec(obs: String, sde: Int, dim: Int, period: Int = 0, harmonics: Int = 0)

  def apply(leaves: Seq[LeafSpec], p: Parameters, precision: Int = 0, df: Int = 0): (Array[Int], Array[Double]) = {
    val nodes: Seq[ParamNode] = p.flatten                      // Tree.flatten, Tree.scala:49-53
    require(nodes.size == leaves.size, "parameter tree shape does not match the composed model")
    val obsKind = leaves.head.obs match { case "poisson" => poisson; case "lgcp" => lgcp; case "negbin" => 3; case "zip" => 4; case "bernoulli" => 5; case "studentt" => 6; case "beta" => 7; case _ => gaussian }
    val ints = Array.newBuilder[Int]; val reals = Array.newBuilder[Double]
    ints ++= Seq(leaves.size, obsKind, precision, df)
    leaves.zip(nodes).foreach { case (l, node) =>
      val (m0, c0, mu, phi, sigma) = node.sdeParam match {     // STORED values: SdeParameters.scala:176-205
        case BrownianParameter(m, c, s)       => (m, c, DenseVector[Double](), DenseVector[Double](), s)
        case GenBrownianParameter(m, c, mu, s) => (m, c, mu, DenseVector[Double](), s)
        case OuParameter(m, c, ph, mu, s)      => (m, c, mu, ph, s)
      }
      ints ++= Seq(l.sde, l.dim, if (l.obs == "seasonal") fSeasonal else fFirst, l.period, l.harmonics,
                   if (node.scale.isDefined) 1 else 0, m0.size, c0.size, mu.size, phi.size, sigma.size)
      reals += node.scale.getOrElse(0.0)
      reals ++= m0.data; reals ++= c0.data; reals ++= mu.data; reals ++= phi.data; reals ++= sigma.data
    }
    (ints.result(), reals.result())
  }
}

/**
  * The GPU particle filter behind the reference's own trait (ParticleFilter.scala:96-167): overrides
  * initialiseState / stepFilter / llFilter / filter wholesale, as FilterLgcp already overrides
  * stepFilter (:210).  The cloud stays in HBM: `PfState.particles` is downloaded only on demand via
  * `currentParticles`; the PfState values returned here carry an empty Vector.
  */
case class FilterGpu(
  leaves: Seq[DescriptorBuilder.LeafSpec],
  params: Parameters,
  mod: Model,
  particles: Int,
  seed: Long = 20260101L,
  device: Int = 0,
  precision: Int = 0) extends ParticleFilter[State] with AutoCloseable {

  private val (ints, reals) = DescriptorBuilder(leaves, params, precision)
  private val handle = CssmNative.create(ints, reals, particles.toLong, seed, device)
  def close(): Unit = CssmNative.destroy(handle)

  // the abstract members exist for source compatibility; the native path never calls them
  def dataLikelihood(g: Gamma, y: Observation): LogLikelihood = mod.dataLikelihood(g, y)
  def stepFunction(dt: TimeIncrement)(s: State) = mod.sde.stepFunction(dt)(s)
  def initialState = mod.sde.initialState
  def f(s: State, t: Time): Gamma = mod.f(s, t)
  def resample: Resample[State] = GpuResample.systematic(device)

  override def initialiseState(n: Int, t0: Time): PfState[State] = {
    require(n == particles, "the handle was created for a fixed particle count")
    CssmNative.init(handle, t0)
    PfState(t0, None, Vector.empty, 0.0, n)
  }

  override def stepFilter(s: PfState[State], y: Data): PfState[State] = {
    val out = new Array[Double](2)
    CssmNative.step(handle, y.t, y.observation.getOrElse(0.0), y.observation.isDefined, out)
    PfState(y.t, y.observation, Vector.empty, out(0), out(1).toInt)
  }

  override def llFilter(data: Vector[Data], n: Int): LogLikelihood =
    CssmNative.filter(handle, data.map(_.t).toArray, data.map(_.observation.getOrElse(0.0)).toArray,
                      data.map(d => (if (d.observation.isDefined) 1 else 0).toByte).toArray, null)

  override def filter(data: Vector[Data], n: Int): (LogLikelihood, Vector[StateSpace[State]]) = {
    val d = mod.sde.dimension
    val path = new Array[Double]((data.size + 1) * d)
    val ll = CssmNative.filter(handle, data.map(_.t).toArray, data.map(_.observation.getOrElse(0.0)).toArray,
                               data.map(x => (if (x.observation.isDefined) 1 else 0).toByte).toArray, path)
    val times = data.minBy(_.t).t +: data.map(_.t)
    (ll, times.zipWithIndex.map { case (t, i) => StateSpace[State](t, Tree.leaf(DenseVector(path.slice(i * d, (i + 1) * d)))) })
  }

  override def filterStream(t0: Time, n: Int): Flow[Data, PfState[State], NotUsed] =
    Flow[Data].scan(initialiseState(n, t0))(stepFilter)        // unchanged: ParticleFilter.scala:163-166

  /** PfState.particles on demand, SoA [d][N]. */
  def currentParticles: Array[Double] = {
    val out = new Array[Double](mod.sde.dimension * particles); CssmNative.particles(handle, out); out
  }

  /** New parameters for the same model: one call per PMMH proposal (PMMH.scala:71). */
  def reparameterise(p: Parameters, newSeed: Long): Unit = {
    val (i, r) = DescriptorBuilder(leaves, p, precision); CssmNative.setParams(handle, i, r, newSeed)
  }
}

/** `Resample[A]` (package.scala:23) on the GPU: ancestor indices via JNI, gather on the JVM.  A parity
  * shim -- it leaves the cloud on the heap; FilterGpu is the production path. */
object GpuResample {
  def systematic[A](device: Int = 0): Resample[A] = (particles: Vector[A], weights: Vector[LogLikelihood]) => {
    val anc = new Array[Int](weights.size)
    CssmNative.resampleSystematic(weights.toArray, scala.util.Random.nextDouble, anc, device)   // u: Resampling.scala:66
    anc.toVector.map(particles(_))
  }
  /** Resampling.stratifiedResampling (:78-86) / multinomialResampling (:92-96): kind 1 / 2 of cssm_resample; the per-slot
    * uniforms come from the library's Philox streams, keyed by a seed drawn where the reference draws its uniforms. */
  private def seeded[A](kind: Int, device: Int): Resample[A] = (particles: Vector[A], weights: Vector[LogLikelihood]) => {
    val anc = new Array[Int](weights.size)
    CssmNative.resample(kind, weights.toArray, 0.0, scala.util.Random.nextLong, 0, anc, device)
    anc.toVector.map(particles(_))
  }
  def stratified[A](device: Int = 0): Resample[A] = seeded(1, device)
  def multinomial[A](device: Int = 0): Resample[A] = seeded(2, device)
}

/** BootstrapFilter for PMMH (package.scala:24; examples/DetermineParameters.scala:70-75): drops into
  * MetropolisHastings.pmmhState(...) unchanged. */
object GpuBootstrap {
  def apply(leaves: Seq[DescriptorBuilder.LeafSpec], unparam: UnparamModel, init: Parameters, data: Vector[Data], n: Int,
            seed: Long = 20260101L): BootstrapFilter[Parameters, StateSpace[State]] = {
    val gpu = FilterGpu(leaves, init, unparam.run(init).get, n, seed)
    var calls = 0L
    Reader { (p: Parameters) => calls += 1; gpu.reparameterise(p, seed + calls); gpu.filter(data, n) }
  }
}
