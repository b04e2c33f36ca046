"""The JNI shim (mvtopicmodel_amd/java/mvhdp_jni.cpp) cannot be built here (no JDK).  What can be checked without one:
it type-checks against the JNI signatures it uses (a declaration-only stub, -fsyntax-only: nothing is built or linked),
it never opens a critical region (every mvhdp_* call may block), and every native method the Java class
declares has its Java_..._n* entry in the shim and vice versa."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "mvtopicmodel_amd", "java", "mvhdp_jni.cpp")
JAVA = os.path.join(ROOT, "mvtopicmodel_amd", "java", "org", "madgik", "MVTopicModel", "NativeSampler.java")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_shim_type_checks_against_the_jni_signatures():
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror",
                           "-I", os.path.join(ROOT, "tests", "native", "jni_stub"), "-I", os.path.join(ROOT, "include"), SHIM])


def test_shim_holds_no_critical_region_and_matches_the_java_class():
    src = re.sub(r"//[^\n]*", "", open(SHIM).read())
    assert "PrimitiveArrayCritical" not in src
    natives = set(re.findall(r"private static native \w[\w\[\]]* (n\w+)\(", open(JAVA).read()))
    entries = set(re.findall(r"Java_org_madgik_MVTopicModel_NativeSampler_(n\w+)\(", src))
    assert natives == entries and len(natives) >= 14
