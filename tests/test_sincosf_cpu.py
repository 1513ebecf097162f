"""include/vo_sincosf.h (the device's cosf/sinf for vo_orb_params.trig_mode = 1) against this host's libm, which is
what the oracle calls for that variant: bit-equal on a strided sweep of every float in [0, 6.3] (the descriptor angle is
degrees in [0, 360) times pi/180) plus the exact angles the ORB path can produce near quadrant boundaries."""
import os
import subprocess
import sys

from conftest import ROOT

SRC = r"""
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "vo_sincosf.h"
int main(int argc, char** argv) {
    long bad = 0, n = 0;
    float hi = 6.3f; unsigned uh; memcpy(&uh, &hi, 4);
    for (unsigned u = 0; u <= uh; u += 61) {
        float x; memcpy(&x, &u, 4);
        float c = cosf(x), s = sinf(x), c2 = vo_glibc_cosf(x), s2 = vo_glibc_sinf(x);
        bad += memcmp(&c, &c2, 4) != 0; bad += memcmp(&s, &s2, 4) != 0; n++;
    }
    for (int d = 0; d < 360 * 4096; d++) {          /* angle (degrees, multiples of 2^-12) * (float)(pi/180) as orb.cpp forms it */
        float a = (float)d / 4096.f; a *= (float)(M_PI / 180.f);
        float c = cosf(a), s = sinf(a), c2 = vo_glibc_cosf(a), s2 = vo_glibc_sinf(a);
        bad += memcmp(&c, &c2, 4) != 0; bad += memcmp(&s, &s2, 4) != 0; n++;
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
"""


def test_restated_sincosf_equals_host_libm(tmp_path):
    c = tmp_path / "t.c"
    c.write_text(SRC)
    exe = str(tmp_path / "t")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(c), "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True)
    n, bad = map(int, out.stdout.split())
    assert n > 15_000_000 and bad == 0, out.stdout
