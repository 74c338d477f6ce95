"""integration/jni/kmcjni.c — the JNI glue BASELINE's north star asks for — cannot be built for real here
(no JDK, no JVM, TLC absent).  What can be checked without them: its C compiles cleanly against a stand-in
jni.h that carries the JNI specification's signatures (tests/jni_stub/jni.h), every native method of
KmcModelChecker.java has its Java_... function, and the object file's undefined symbols are exactly entry
points that include/kmc.h declares and libkmc.so exports (plus libc)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "integration", "jni", "kmcjni.c")
JAVA = os.path.join(ROOT, "integration", "jni", "KmcModelChecker.java")


def test_jni_glue_compiles_and_binds_only_the_c_abi(tmp_path):
    obj = tmp_path / "kmcjni.o"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fPIC", "-c", SRC, "-o", str(obj),
                           "-I" + os.path.join(ROOT, "tests", "jni_stub"), "-I" + os.path.join(ROOT, "include")])
    syms = subprocess.run(["nm", str(obj)], capture_output=True, text=True, check=True).stdout.splitlines()
    undefined = {l.split()[-1] for l in syms if " U " in l}
    defined = {l.split()[-1] for l in syms if " T " in l}
    header = open(os.path.join(ROOT, "include", "kmc.h")).read()
    declared = set(re.findall(r"\b(kmc_[a-z_0-9]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    kmc_used = {s for s in undefined if s.startswith("kmc_")}
    assert kmc_used and kmc_used <= declared, kmc_used - declared
    assert {s for s in undefined if not s.startswith("kmc_")} <= {"malloc", "free", "strlen", "strncpy", "strncat",
                                                                 "memset", "__stack_chk_fail", "_GLOBAL_OFFSET_TABLE_"}
    # every `native` method of the Java half has its exported C function, and nothing else is exported
    natives = set(re.findall(r"public static native [\w\[\]]+ (\w+)\(", open(JAVA).read()))
    assert natives == {"open", "run", "result", "trace", "contains", "checkpoint", "recover", "close"}
    assert defined == {"Java_tlc2_tool_gpu_KmcModelChecker_" + n for n in natives}
    lib = subprocess.run(["nm", "-D", os.path.join(ROOT, "kafka_specification_amd", "libkmc.so")],
                         capture_output=True, text=True, check=True).stdout
    for s in kmc_used:
        assert re.search(rf" T {s}$", lib, flags=re.M), f"libkmc.so does not export {s}"


HARNESS = os.path.join(ROOT, "tests", "_jni_harness")


def build_harness():
    """tests/jni_stub/fake_jvm.c (a JNIEnv implemented over a toy object model + a C main playing the Java half) linked with
    the real kmcjni.c and libkmc.so: every line of the glue executes — against a stand-in, not a JVM."""
    src = [os.path.join(ROOT, "tests", "jni_stub", "fake_jvm.c"), SRC]
    lib = os.path.join(ROOT, "kafka_specification_amd")
    if not os.path.exists(HARNESS) or any(os.path.getmtime(HARNESS) < os.path.getmtime(s) for s in src):
        subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                               "-I" + os.path.join(ROOT, "include"), *src, "-L" + lib, "-lkmc", "-Wl,-rpath," + lib,
                               "-o", HARNESS])
    return HARNESS


def test_jni_glue_executes_and_turns_a_failed_open_into_a_java_exception():
    """No GPU here: kmc_open fails ("no HIP device ... no CPU fallback"), and the glue must hand that to the caller as an
    IllegalStateException carrying kmc_last_error() — after marshalling all seventeen Config fields through GetFieldID /
    Get<Type>Field without an exception of its own (a misspelt field would surface as NoSuchFieldError instead)."""
    import json
    env = dict(os.environ, KMC_NO_TORCH="1", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    p = subprocess.run([build_harness(), "5", "3", "2", "2", "1", "7", "device=0"], capture_output=True, text=True, env=env, timeout=120)
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 3 and out["exception"] == "java/lang/IllegalStateException"
    assert out["message"].startswith("kmc_open: ") and "no CPU fallback" in out["message"]
