"""include/yolo355.h is the product boundary: it must be valid plain C and a C program must link against libyolo355.so.
Builds tests/c_client/client.c with gcc (C99, warnings as errors) and runs it -- argument checks only, no GPU needed."""
import os
import subprocess

from yolo355 import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    exe = str(tmp_path / "client")
    libdir = os.path.dirname(_ffi.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_client", "client.c"), "-o", exe, "-L", libdir, "-l:libyolo355.so",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("ok version")


def test_header_compiles_as_cxx_too(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "yolo355.h"\nint main() { return y355_version() < 0; }\n')
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
