"""Process-wide knobs of the HIP path."""
import os

# "bf16x3": operands carry hi + lo planes, 3 MFMAs per product -- fp16 planes since r03 (~2^-22 relative error; the keyword is
#           kept from the bf16-pair rounds, ~2^-17), fp32-class: the mode the parity tests pin against the fp32 oracle;
# "bf16":   single bf16 plane, 1 MFMA per product (the throughput mode BASELINE.json config 2 names).
PRECISION = os.environ.get("FRIDO_PRECISION", "bf16x3")


# "bf16x3_bf16" (r05): the same two-plane arithmetic on the library build whose planes are bf16 pairs (fp32's range, 2^-17 relative):
#           for checkpoints whose un-normalised operands leave fp16's +-65504 (the status word's FRIDO_STATUS_SATURATED says so).


# (r06) WHICH plane format a two-plane model runs on is the library's choice, not the user's: a module whose precision is the default
#           keyword "bf16x3" starts on the fp16 pairs (22 mantissa bits) and, when a run saturates an fp16 operand plane (the status word's
#           FRIDO_STATUS_SATURATED, polled in stream order after every sampling pass / decode / forward), is moved to the bf16-pair build
#           and the run is repeated there (frido_amd/autoplanes.py).  "bf16x3_f16" pins the fp16 pairs (saturation then only warns),
#           "bf16x3_bf16" pins the bf16 pairs; FRIDO_AUTO_PLANES=0 turns the automatic move off process-wide.
AUTO_PLANES = os.environ.get("FRIDO_AUTO_PLANES", "1") != "0"


def nsplit(precision=None):
    p = precision or PRECISION
    if p in ("bf16x3", "bf16x3_bf16", "bf16x3_f16"):
        return 2
    if p == "bf16":
        return 1
    raise ValueError(f"unknown precision '{p}' (use 'bf16x3', 'bf16x3_f16', 'bf16x3_bf16' or 'bf16')")


def auto_planes(precision=None):
    """True when the plane format of a module with this precision keyword is the library's to choose (see AUTO_PLANES above)."""
    return AUTO_PLANES and (precision or PRECISION) == "bf16x3"


def planes(precision=None):
    """Two-plane element format the precision keyword selects: which build of the library the model runs on (_lib.use_planes)."""
    return "bf16" if (precision or PRECISION) == "bf16x3_bf16" else "f16"
