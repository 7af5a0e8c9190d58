"""What a struct-level binding pays per SpendProof before the engine sees a byte: the crate's structs hold
RistrettoPoints, the C ABI takes their 32-byte encodings, so `SpendProof::write_record` runs 130 `compress()` and `from_record` 130
`decompress()` -- one inverse square root each.  Timed here with the repository's own host build of the decode / encode routines
(tests/hostcheck: the device headers compiled by g++, portable C, 9 x 29-bit limbs) on ONE core of this box.  dalek's 64-bit serial
backend (5 x 51-bit limbs) does the same field work in fewer multiplications, so read the result as an order of magnitude, not as
dalek's number.  No GPU needed."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    hc = os.path.join(ROOT, "tests", "hostcheck")
    so = os.path.join(hc, "libhostcheck.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, os.path.join(hc, "hostcheck.cpp")], check=True)
    lib = C.CDLL(so)
    count, reps = 130, 200
    pts = b""
    for i in range(count):
        out = C.create_string_buffer(32)
        lib.hc_from_uniform(hashlib.shake_256(b"marshal-bound-%d" % i).digest(64), out)
        pts += out.raw
    sd, se = C.c_double(0), C.c_double(0); chk = C.create_string_buffer(32)
    assert lib.hc_time_point_codec(pts, count, reps, C.byref(sd), C.byref(se), chk) == 1
    us_d, us_e = 1e6 * sd.value / (count * reps), 1e6 * se.value / (count * reps)
    model = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"
    print("host: %s, one core" % model)
    print("decode (decompress): %.2f us per point   encode (compress): %.2f us per point" % (us_d, us_e))
    print("SpendProof::write_record (130 compress):   %.3f ms per proof = %7.0f proofs/s per core" % (130 * us_e / 1e3, 1e6 / (130 * us_e)))
    print("SpendProof::from_record  (130 decompress): %.3f ms per proof = %7.0f proofs/s per core" % (130 * us_d / 1e3, 1e6 / (130 * us_d)))
    print("cores needed in front of ONE GPU verifying 514 000 proofs/s (write_record only): %.0f" % (514000 * 130 * us_e / 1e6))


if __name__ == "__main__":
    main()
