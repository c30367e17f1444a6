"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU involved): the wire format's validating
reader / writer and the sampler table (tfhe_amd/csrc/gsw_wire.cpp, which parses untrusted bytes), and the CPU
oracle itself.  Each is compiled with gcc's sanitizers into a driver under a temporary directory and run; the
drivers exercise truncated, oversized, bit-flipped and count-overflowing objects on exact-size heap buffers."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
ENV.pop("LD_PRELOAD", None)  # ASan must come first in the initial library list


def _run(binary):
    proc = subprocess.run([binary], env=ENV, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    assert proc.stdout.strip().endswith(": ok"), proc.stdout[-500:]


def test_wire_format_reader_under_sanitizers(tmp_path):
    out = str(tmp_path / "wire_sanitized")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra"] + SAN +
                          ["-I", os.path.join(ROOT, "tfhe_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "wire_sanitized.cpp"),
                           os.path.join(ROOT, "tfhe_amd", "csrc", "gsw_wire.cpp"), "-o", out])
    _run(out)


def test_oracle_under_sanitizers(tmp_path):
    out = str(tmp_path / "oracle_sanitized")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-fopenmp"] + SAN +
                          ["-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "oracle_sanitized.c"),
                           os.path.join(ROOT, "oracle", "gsw_oracle.c"), "-lm", "-o", out])
    _run(out)
