"""Number formatting across the reference's XML-RPC boundary.

env -> server: Java `String.valueOf(float)` (clientCFD.pde:120): shortest decimal that round-trips
the float32, printed as d.ddd for 1e-3 <= |x| < 1e7 and as d.dddE-n otherwise, always with at
least one fractional digit.  server -> env: `Float.parseFloat` (clientCFD.pde:125-126), i.e. the
decimal string rounded to float32.
"""
import numpy as np


def java_float_to_string(x):
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "Infinity" if x > 0 else "-Infinity"
    if x == 0:
        return "-0.0" if np.signbit(x) else "0.0"
    s = np.format_float_scientific(x, unique=True, trim="-", exp_digits=1)  # e.g. '8.495961e-3', '-1.e+0'
    mant, exp = s.split("e")
    exp = int(exp)
    neg = mant.startswith("-")
    mant = mant.lstrip("-")
    digits = mant.replace(".", "")
    if digits == "":
        digits = "0"
    ax = abs(float(x))
    if 1e-3 <= ax < 1e7:
        # plain decimal: place the point after (exp+1) digits
        if exp >= 0:
            ip = digits[:exp + 1].ljust(exp + 1, "0")
            fp = digits[exp + 1:] or "0"
        else:
            ip = "0"
            fp = "0" * (-exp - 1) + digits
        out = ip + "." + fp
    else:
        out = digits[0] + "." + (digits[1:] or "0") + "E" + str(exp)
    return ("-" if neg else "") + out


def java_parse_float(s):
    return np.float32(float(s))
