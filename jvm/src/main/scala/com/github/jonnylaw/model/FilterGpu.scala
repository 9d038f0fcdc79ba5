package com.github.jonnylaw.model

// The Scala side of the drop-in boundary (no Scala compiler in the build image: source; the JNI layer below it is compiled
// and driven wherever a JDK exists -- tests/test_jvm_binding.py).
// It lives in package com.github.jonnylaw.model because the model case classes it pattern-matches
// (PoissonModel, SeasonalModel, LinearModel, LogGaussianCox, BrownianMotion, GenBrownianMotion,
// OuProcess) are `private final` there (Model.scala:144-363, Sde.scala:69,98,129).

import akka.NotUsed
import akka.stream.scaladsl.Flow
import breeze.linalg.DenseVector
import cats.data.Reader

// CssmNative -- the JNI entry points -- is a Java class with static natives (jvm/src/main/java/.../CssmNative.java): the glue's
// symbols take a jclass, which a Scala `object` with @native methods (natives on the module class CssmNative$) would not match.

/** Flattens a parameterised composed model into the cssm_model_desc wire form (see cssm_jni.c). */
object DescriptorBuilder {
  private val (brownian, genBrownian, ou) = (0, 1, 2)
  private val (fFirst, fSeasonal) = (0, 1)
  private val (poisson, gaussian, lgcp) = (0, 1, 2)

  /** One leaf of the composition, left to right: (obs/f kind, sde kind, dimension, period, harmonics).  Users never write
    * these: they come from LeafSpecs.ofModel (a single leaf model, by matching the private case classes) or from the
    * structure-carrying constructors of GpuModel (a composition). */
  case class LeafSpec(obs: String, sde: Int, dim: Int, period: Int = 0, harmonics: Int = 0, df: Int = 0)

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

/** The leaf structure of a parameterised model, recovered inside package `model` where the case classes are visible
  * (Model.scala:144,168,204,241,266,281,315,339,363; Sde.scala:69,98,129). */
object LeafSpecs {
  import DescriptorBuilder.LeafSpec
  def ofSde(s: Sde): (Int, Int) = s match {
    case BrownianMotion(_, d)    => (0, d)
    case GenBrownianMotion(_, d) => (1, d)
    case OuProcess(_, d)         => (2, d)
    case other => throw new IllegalArgumentException(s"the GPU filter serves the built-in SDEs; got $other")
  }
  /** Some(leaf) for a single-leaf model.  A COMPOSED model is an anonymous `new Model { ... }` closure (Model.compose,
    * Model.scala:110-136): its leaves cannot be recovered from the value -- compose with GpuModel instead, which
    * carries the structure alongside the reference's own UnparamModel. */
  def ofModel(m: Model): Option[Seq[LeafSpec]] = {
    def leaf(obs: String, sde: Sde, period: Int = 0, harmonics: Int = 0, df: Int = 0) = {
      val (k, d) = ofSde(sde); Some(Seq(LeafSpec(obs, k, d, period, harmonics, df)))
    }
    m match {
      case PoissonModel(sde, _)             => leaf("poisson", sde)
      case SeasonalModel(p, h, sde, _)      => leaf("seasonal", sde, p, h)
      case LinearModel(sde, _)              => leaf("linear", sde)
      case StudentsTModel(sde, df, _)       => leaf("studentt", sde, df = df)
      case NegativeBinomialModel(sde, _)    => leaf("negbin", sde)
      case ZeroInflatedPoisson(sde, _)      => leaf("zip", sde)
      case BernoulliModel(sde, _)           => leaf("bernoulli", sde)
      case BetaModel(sde, _)                => leaf("beta", sde)
      case LogGaussianCox(sde, _)           => leaf("lgcp", sde)
      case _                                => None
    }
  }
}

/** Composition that keeps its structure: the reference's constructors (Model.poisson, Model.seasonal, ... Model.scala:44-92)
  * and `|+|` (Model.scala:96-100), with the leaf list carried beside the UnparamModel they build.  A call site changes
  * one identifier:  `GpuModel.poisson(Sde.ouProcess(1)) |+| GpuModel.seasonal(24, 1, Sde.ouProcess(2))`. */
case class GpuUnparam(unparam: UnparamModel, leaves: Seq[DescriptorBuilder.LeafSpec]) {
  def |+|(that: GpuUnparam): GpuUnparam = GpuUnparam(Model.compose(unparam, that.unparam), leaves ++ that.leaves)
}
object GpuModel {
  import DescriptorBuilder.LeafSpec
  /** kind and dimension of an unparameterised SDE: Sde.brownianMotion(d) etc. are Readers of the parameter leaf, so the
    * structure is read off a run with a neutral parameter of the matching constructor. */
  private def kindOf(sde: UnparamSde): (Int, Int) = {
    val z = breeze.linalg.DenseVector(0.0)
    val probes = Seq[SdeParameter](BrownianParameter(z, z, z), GenBrownianParameter(z, z, z, z), OuParameter(z, z, z, z, z))
    // (the reference's mismatch branch is `Failure(throw new Exception(...))` -- Sde.scala:183,188,201: the throw is evaluated
    //  eagerly, so run() THROWS on a parameter of another SDE instead of returning the Failure; Try catches it)
    probes.view.flatMap(p => scala.util.Try(sde.run(p)).flatten.toOption).headOption.map(LeafSpecs.ofSde)
      .getOrElse(throw new IllegalArgumentException("the GPU filter serves the built-in SDEs"))
  }
  private def leaf(obs: String, u: UnparamModel, sde: UnparamSde, period: Int = 0, harmonics: Int = 0, df: Int = 0) = {
    val (k, d) = kindOf(sde); GpuUnparam(u, Seq(LeafSpec(obs, k, d, period, harmonics, df)))
  }
  def poisson(sde: UnparamSde)                       = leaf("poisson", Model.poisson(sde), sde)
  def seasonal(period: Int, h: Int, sde: UnparamSde) = leaf("seasonal", Model.seasonal(period, h, sde), sde, period, h)
  def linear(sde: UnparamSde)                        = leaf("linear", Model.linear(sde), sde)
  def studentsT(sde: UnparamSde, df: Int)            = leaf("studentt", Model.studentsT(sde, df), sde, df = df)
  def negativeBinomial(sde: UnparamSde)              = leaf("negbin", Model.negativeBinomial(sde), sde)
  def zeroInflatedPoisson(sde: UnparamSde)           = leaf("zip", Model.zeroInflatedPoisson(sde), sde)
  def bernoulli(sde: UnparamSde)                     = leaf("bernoulli", Model.bernoulli(sde), sde)
  def beta(sde: UnparamSde)                          = leaf("beta", Model.beta(sde), sde)
  def lgcp(sde: UnparamSde)                          = leaf("lgcp", Model.lgcp(sde), sde)
}

/** `Resample[A]` (package.scala:23) values the native filter understands: as a function they are the parity shim (ancestor
  * indices via JNI, gather on the JVM); handed to FilterGpu they only select the resampler of the device path
  * (CSSM_OPT_RESAMPLER) -- the cloud never leaves HBM.  Resampling.scala:63-72 (systematic), :78-86, :92-96. */
sealed abstract class GpuResampler[A](val kind: Int, device: Int) extends ((Vector[A], Vector[LogLikelihood]) => Vector[A]) {
  def apply(particles: Vector[A], weights: Vector[LogLikelihood]): Vector[A] = {
    val anc = new Array[Int](weights.size)
    if (kind == 0) CssmNative.resampleSystematic(weights.toArray, scala.util.Random.nextDouble, anc, device)   // u: Resampling.scala:66
    else CssmNative.resample(kind, weights.toArray, 0.0, scala.util.Random.nextLong, 0, anc, device)           // keyed where the reference draws
    anc.toVector.map(particles(_))
  }
}
object GpuResample {
  def systematic[A](device: Int = 0): GpuResampler[A]  = new GpuResampler[A](0, device) {}
  def stratified[A](device: Int = 0): GpuResampler[A]  = new GpuResampler[A](1, device) {}
  def multinomial[A](device: Int = 0): GpuResampler[A] = new GpuResampler[A](2, device) {}
}

/**
  * The GPU particle filter behind the reference's own trait (ParticleFilter.scala:96-167), with the constructor shape of
  * `Filter(mod, resample)` (:233-235): overrides initialiseState / stepFilter / llFilter / filter wholesale, as FilterLgcp
  * already overrides stepFilter (:210).  The cloud stays in HBM: `PfState.particles` is downloaded only on demand via
  * `currentParticles`; the PfState values returned here carry an empty Vector.
  *
  * `leaves`: for a single-leaf model they are derived from `mod` itself; for a composition pass `gpuUnparam.leaves`
  * (see GpuModel) -- FilterGpu.of does both.
  */
case class FilterGpu(
  mod: Model,
  resample: Resample[State],
  params: Parameters,
  particles: Int,
  leaves: Seq[DescriptorBuilder.LeafSpec] = Nil,
  seed: Long = 20260101L,
  device: Int = 0,
  precision: Int = 0) extends ParticleFilter[State] with AutoCloseable {

  private val structure: Seq[DescriptorBuilder.LeafSpec] =
    if (leaves.nonEmpty) leaves
    else LeafSpecs.ofModel(mod).getOrElse(throw new IllegalArgumentException(
      "a composed model does not reveal its leaves (Model.compose builds a closure): compose with GpuModel and pass its `leaves`"))
  private val df = structure.head.df                                     // StudentsTModel.df (Model.scala:144)
  private val resamplerKind = resample match {
    case g: GpuResampler[_] => g.kind
    case _ => throw new IllegalArgumentException("the device path resamples with GpuResample.systematic / stratified / multinomial")
  }
  private val (ints, reals) = DescriptorBuilder(structure, params, precision, df)
  private val handle = CssmNative.create(ints, reals, particles.toLong, seed, device)
  if (resamplerKind != 0) CssmNative.setResampler(handle, resamplerKind)
  /** the handle's clock: the time of the last state it produced (the handle is mutable, PfState values are not) */
  private var clock: Double = Double.NaN
  def close(): Unit = CssmNative.destroy(handle)

  // the abstract members exist for source compatibility; the native path never calls them
  def dataLikelihood(g: Gamma, y: Observation): LogLikelihood = mod.dataLikelihood(g, y)
  def stepFunction(dt: TimeIncrement)(s: State) = mod.sde.stepFunction(dt)(s)
  def initialState = mod.sde.initialState
  def f(s: State, t: Time): Gamma = mod.f(s, t)

  override def initialiseState(n: Int, t0: Time): PfState[State] = {
    require(n == particles, "the handle was created for a fixed particle count")
    CssmNative.init(handle, t0)
    clock = t0
    PfState(t0, None, Vector.empty, 0.0, n)
  }

  override def stepFilter(s: PfState[State], y: Data): PfState[State] = {
    // the state argument is a value, the cloud behind it is the handle's: a replayed or branched stream that hands in an
    // OLDER state would otherwise be stepped from the wrong cloud without a word
    if (s.t != clock) throw new IllegalStateException(
      s"stepFilter on a state of time ${s.t}, but the device cloud is at $clock: a FilterGpu advances one stream; re-initialise to branch")
    val out = new Array[Double](2)
    CssmNative.step(handle, y.t, y.observation.getOrElse(0.0), y.observation.isDefined, out)
    clock = y.t
    PfState(y.t, y.observation, Vector.empty, out(0), out(1).toInt)
  }

  override def llFilter(data: Vector[Data], n: Int): LogLikelihood = {
    val ll = CssmNative.filter(handle, data.map(_.t).toArray, data.map(_.observation.getOrElse(0.0)).toArray,
                               data.map(d => (if (d.observation.isDefined) 1 else 0).toByte).toArray, null)
    clock = data.last.t
    ll
  }

  override def filter(data: Vector[Data], n: Int): (LogLikelihood, Vector[StateSpace[State]]) = {
    val d = mod.sde.dimension
    val path = new Array[Double]((data.size + 1) * d)
    val ll = CssmNative.filter(handle, data.map(_.t).toArray, data.map(_.observation.getOrElse(0.0)).toArray,
                               data.map(x => (if (x.observation.isDefined) 1 else 0).toByte).toArray, path)
    clock = data.last.t
    val times = data.minBy(_.t).t +: data.map(_.t)
    (ll, times.zipWithIndex.map { case (t, i) => StateSpace[State](t, toState(path, i * d)) })
  }

  /** One device row (d doubles, Tree.flatten order) as the State the reference's own SDE would hold: one Leaf per composed
    * sub-model, LEFT-nested -- `(a |+| b) |+| c` is Branch(Branch(a, b), c) (Tree.scala:18-20; Sde.scala:206-209 builds the
    * composed initial state that way).  A flat Leaf would type-check and be WRONG: Model.compose's `f` on a Leaf evaluates
    * model1.f only (Model.scala:122-128), silently dropping every other leaf's contribution. */
  private def toState(row: Array[Double], off: Int): State = {
    var o = off
    val leaves = structure.map { l => val v = DenseVector(row.slice(o, o + l.dim)); o += l.dim; Tree.leaf(v) }
    leaves.reduceLeft((acc, leaf) => Tree.branch(acc, leaf))
  }

  /** The current cloud as the reference's `PfState.particles` would hold it (Vector[State], length N) -- a download of
    * d x N doubles: for inspection and for consumers of `mod.f(state, t)`, not for the per-step path. */
  def currentStates: Vector[State] = {
    val soa = currentParticles; val d = mod.sde.dimension
    Vector.tabulate(particles) { i => val row = Array.tabulate(d)(k => soa(k * particles + i)); toState(row, 0) }
  }

  override def filterStream(t0: Time, n: Int): Flow[Data, PfState[State], NotUsed] =
    Flow[Data].scan(initialiseState(n, t0))(stepFilter)        // unchanged: ParticleFilter.scala:163-166

  /** PfState.particles on demand, SoA [d][N]. */
  def currentParticles: Array[Double] = {
    val out = new Array[Double](mod.sde.dimension * particles); CssmNative.particles(handle, out); out
  }

  /** New parameters for the same model: one call per PMMH proposal (PMMH.scala:71).  `key`: CssmNative.runKey(seed, run). */
  def reparameterise(p: Parameters, key: Long): Unit = {
    val (i, r) = DescriptorBuilder(structure, p, precision, df); CssmNative.setParams(handle, i, r, key)
  }
}

object FilterGpu {
  /** `Filter(unparam.run(p).get, resample)` for a structure-carrying composition. */
  def of(g: GpuUnparam, p: Parameters, resample: Resample[State], particles: Int, seed: Long = 20260101L, device: Int = 0,
         precision: Int = 0): FilterGpu =
    FilterGpu(g.unparam.run(p).get, resample, p, particles, g.leaves, seed, device, precision)
}

/** BootstrapFilter for PMMH (package.scala:24; examples/DetermineParameters.scala:70-75): drops into
  * MetropolisHastings.pmmhState(...) unchanged. */
object GpuBootstrap {
  def apply(g: GpuUnparam, init: Parameters, data: Vector[Data], n: Int,
            seed: Long = 20260101L): BootstrapFilter[Parameters, StateSpace[State]] = {
    val gpu = FilterGpu.of(g, init, GpuResample.systematic[State](), n, seed)
    var calls = 0L
    // every filter run gets its own Philox key, a PRF of (seed, run) -- never seed + run: chains seeded s and s + 1 would
    // replay each other's filter randomness one iteration apart
    Reader { (p: Parameters) => calls += 1; gpu.reparameterise(p, CssmNative.runKey(seed, calls)); gpu.filter(data, n) }
  }
}
