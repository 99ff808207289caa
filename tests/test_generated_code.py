"""CPU checks of the generated arithmetic headers: the committed files are what the generators emit, and the windowed
table product (csrc/fr29.hpp: f29_mulw) is exact against Python integers at the stated operand bounds."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen(script, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], check=True, capture_output=True, text=True).stdout


def test_generated_headers_are_current():
    for script, args, header in (("gen_fr29_mulw.py", (), "fr29_mulw_gen.hpp"), ("gen_fr29_mulw.py", ("--uniform",), "fr29_mulw_s_gen.hpp"),
                                 ("gen_fr29_montmul.py", (), "fr29_montmul_gen.hpp")):
        with open(os.path.join(ROOT, "ligero-prover_amd", "csrc", header)) as f:
            assert f.read() == _gen(script, *args), header


def test_windowed_product_is_exact_and_fits_64_bit_columns():
    assert _gen("gen_fr29_mulw.py", "--check").startswith("ok:")
