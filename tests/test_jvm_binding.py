"""The JVM side of the drop-in boundary (jvm/): CssmNative.java + cssm_jni.c + the Scala shim FilterGpu.scala.

CPU part: the three sources agree with each other and with include/cssm_pf.h (names, the State tree the shim builds).
GPU part (-m gpu): wherever a JDK exists on the box, compile the JNI glue and a small Java harness, drive
create -> init -> step -> particles -> filter -> llFilter through JNI and compare with the same calls through ctypes,
bit for bit; where there is none, skip WITH the probe's findings ("probe for java on the GPU box, do not assume").
"""
import glob
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JVM = os.path.join(ROOT, "jvm", "src")
GLUE = os.path.join(JVM, "main", "c", "cssm_jni.c")
NATIVE = os.path.join(JVM, "main", "java", "com", "github", "jonnylaw", "model", "CssmNative.java")
SHIM = os.path.join(JVM, "main", "scala", "com", "github", "jonnylaw", "model", "FilterGpu.scala")
SMOKE = os.path.join(JVM, "test", "java", "CssmJniSmoke.java")


def test_jni_glue_java_natives_and_scala_shim_name_the_same_entry_points():
    glue, native, shim = open(GLUE).read(), open(NATIVE).read(), open(SHIM).read()
    c_names = set(re.findall(r"Java_com_github_jonnylaw_model_CssmNative_(\w+)\(JNIEnv\* env, jclass", glue))
    j_names = set(re.findall(r"public static native \S+ (\w+)\(", native))
    assert c_names and c_names == j_names, (sorted(c_names ^ j_names))
    used = set(re.findall(r"CssmNative\.(\w+)\(", shim))
    assert used and used <= j_names, sorted(used - j_names)
    # the natives live on a Java class with static methods: a Scala `object` would put them on CssmNative$ (another JNI name)
    assert "object CssmNative" not in shim
    # every C-ABI function the glue calls is declared in the public header
    header = open(os.path.join(ROOT, "include", "cssm_pf.h")).read()
    for fn in set(re.findall(r"\b(cssm_\w+)\(", glue)):
        assert re.search(r"\b%s\(" % fn, header), fn


def test_scala_shim_builds_left_nested_states_and_survives_the_reference_throwing_readers():
    shim = open(SHIM).read()
    # State of a composed model = Branch(Branch(a, b), c) (Tree.scala:18-20, Sde.scala:206-209): never one flat Leaf per row
    assert "Tree.leaf(DenseVector(path.slice" not in shim
    assert re.search(r"reduceLeft\(\(acc, leaf\) => Tree\.branch\(acc, leaf\)\)", shim)
    assert "StateSpace[State](t, toState(path, i * d))" in shim
    # Sde.brownianMotion(d).run(p) THROWS for a parameter of another SDE (Failure(throw ...), Sde.scala:183,188,201)
    assert "scala.util.Try(sde.run(p)).flatten.toOption" in shim
    assert "sde.run(p).toOption" not in shim.replace("scala.util.Try(sde.run(p)).flatten.toOption", "")


def test_jni_glue_passes_a_syntax_only_compile():
    """gcc -fsyntax-only of the glue against tests/jni_stub/jni.h -- a TEST-ONLY prototype header written from the public JNI
    specification (only the members the glue calls): catches arity and type slips in the calls through JNIEnv and into the C ABI.
    Hygiene, not evidence of a binding: nothing compiled against the stub is linked or run."""
    res = subprocess.run(["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                          "-I" + os.path.join(ROOT, "tests", "jni_stub"), "-I" + os.path.join(ROOT, "include"), GLUE],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]


_JNI_TYPE = {"void": "void", "long": "jlong", "int": "jint", "double": "jdouble", "boolean": "jboolean", "byte": "jbyte",
             "int[]": "jintArray", "double[]": "jdoubleArray", "byte[]": "jbyteArray", "long[]": "jlongArray"}


def test_every_java_native_has_a_glue_function_of_the_matching_jni_signature():
    """`public static native R name(A a, ...)` <-> `JNIEXPORT jR JNICALL Java_..._name(JNIEnv*, jclass, jA, ...)`: return type, arity and
    every argument type (JNI specification, table 3-1 and section 3.2) -- a mismatch links and then corrupts arguments at run time."""
    glue, native = open(GLUE).read(), open(NATIVE).read()
    java = {m.group(2): (m.group(1), [a.strip().rsplit(" ", 1)[0] for a in m.group(3).split(",") if a.strip()])
            for m in re.finditer(r"public static native (\S+) (\w+)\(([^)]*)\);", native)}
    cfun = {m.group(2): (m.group(1), [a.strip() for a in m.group(3).split(",")])
            for m in re.finditer(r"JNIEXPORT (\w+) JNICALL Java_com_github_jonnylaw_model_CssmNative_(\w+)\(([^)]*)\)", glue)}
    assert java and set(java) == set(cfun)
    for name, (ret, args) in java.items():
        cret, cargs = cfun[name]
        assert cret == _JNI_TYPE[ret], (name, ret, cret)
        assert cargs[0].startswith("JNIEnv*") and cargs[1].startswith("jclass"), (name, cargs[:2])
        ctypes_ = [a.rsplit(" ", 1)[0] for a in cargs[2:]]
        assert ctypes_ == [_JNI_TYPE[a] for a in args], (name, args, ctypes_)


def _probe_jdk():
    """(javac, java, include dir holding jni.h) or a string saying what is missing."""
    found = {}
    for tool in ("javac", "java"):
        p = shutil.which(tool)
        if p is None and os.environ.get("JAVA_HOME"):
            q = os.path.join(os.environ["JAVA_HOME"], "bin", tool)
            p = q if os.path.exists(q) else None
        found[tool] = p
    homes = [os.environ.get("JAVA_HOME")] if os.environ.get("JAVA_HOME") else []
    if found["javac"]:
        homes.append(os.path.dirname(os.path.dirname(os.path.realpath(found["javac"]))))
    homes += sorted(glob.glob("/usr/lib/jvm/*")) + sorted(glob.glob("/opt/*jdk*")) + sorted(glob.glob("/opt/java/*"))
    inc = next((os.path.join(h, "include") for h in homes if h and os.path.exists(os.path.join(h, "include", "jni.h"))), None)
    if found["javac"] and found["java"] and inc:
        return found["javac"], found["java"], inc
    return "javac=%s java=%s jni.h=%s (looked in JAVA_HOME=%s, /usr/lib/jvm, /opt)" % (
        found["javac"], found["java"], inc, os.environ.get("JAVA_HOME"))


@pytest.mark.gpu
def test_jni_binding_drives_the_filter_like_the_ctypes_binding(tmp_path):
    probe = _probe_jdk()
    if isinstance(probe, str):
        pytest.skip("no JDK on this box, the JNI layer stays source-only here: " + probe)
    javac, java, inc = probe
    csrc = os.path.join(ROOT, "composablestatespacemodels_amd", "csrc")
    out = str(tmp_path)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-I" + inc, "-I" + os.path.join(inc, "linux"),
                           "-I" + os.path.join(ROOT, "include"), GLUE, "-L" + csrc, "-lcssm_pf", "-Wl,-rpath," + csrc,
                           "-o", os.path.join(out, "libcssm_jni.so")])
    subprocess.check_call([javac, "-d", out, NATIVE, SMOKE])
    n, seed = 4096, 20260101
    res = subprocess.run([java, "-Djava.library.path=" + out, "-cp", out, "CssmJniSmoke", str(n), str(seed)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = dict(ln.split(" ", 1) for ln in res.stdout.strip().splitlines())

    import cases
    from composablestatespacemodels_amd import _abi
    from composablestatespacemodels_amd.filter import NativePf
    t = np.array([1.0, 2.0, 3.0, 4.5, 5.0]); y = np.array([2.0, 0.0, 3.0, 1.0, 4.0]); has = np.array([1, 1, 0, 1, 1], dtype=np.uint8)
    g = NativePf(cases.c1_model(), n, seed)
    g.init(0.0)
    steps = [g.step(t[s], y[s], bool(has[s])) for s in range(3)]
    cloud = g.particles()
    ll_f, _, _, path = g.run(t, y, has, want_path=True)
    ll_l = g.run(t, y, has)[0]
    hx = float.hex

    def same(java_hex, value):   # Double.toHexString and float.hex spell the same double differently: compare the values
        return float.fromhex(java_hex) == value

    jsteps = [ln.split(" ", 1)[1] for ln in res.stdout.strip().splitlines() if ln.startswith("step ")]
    assert len(jsteps) == 3
    for js, (ll, ess) in zip(jsteps, steps):
        a, b = js.split()
        assert same(a, ll) and int(b) == ess, (js, hx(ll), ess)
    c0, c1 = lines["cloud"].split()
    assert same(c0, cloud[0, 0]) and same(c1, cloud[0, n - 1])
    f0, f1 = lines["filter"].split()
    assert same(f0, ll_f) and same(f1, path[len(t), 0])
    assert same(lines["llFilter"], ll_l)
    assert int(lines["runKey"]) == int(_abi.load_library().cssm_pf_run_key(seed, 7))
    # a malformed descriptor surfaces as a RuntimeException carrying cssm_last_error()
    assert lines["error"].startswith("n_leaves"), lines["error"]
    g.close()
