"""Process-wide knobs of the HIP path."""
import os

# "bf16x3": operands carry hi + lo planes, 3 MFMAs per product -- fp16 planes since r03 (~2^-22 relative error; the keyword is
#           kept from the bf16-pair rounds, ~2^-17), fp32-class: the mode the parity tests pin against the fp32 oracle;
# "bf16":   single bf16 plane, 1 MFMA per product (the throughput mode BASELINE.json config 2 names).
PRECISION = os.environ.get("FRIDO_PRECISION", "bf16x3")


def nsplit(precision=None):
    p = precision or PRECISION
    if p == "bf16x3":
        return 2
    if p == "bf16":
        return 1
    raise ValueError(f"unknown precision '{p}' (use 'bf16x3' or 'bf16')")
