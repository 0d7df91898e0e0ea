"""TEST INFRASTRUCTURE ONLY -- import the reference (`/root/reference`, read-only) in THIS container.

Puts the stand-in `simpy` / `gym` packages of this directory in front of the reference on
sys.path, disables bytecode writing (the reference tree is read-only) and chdirs into the
reference root because its scenario paths are relative.  Never used on the GPU box
(`/root/reference` does not exist there) and never imported by the product package.
"""
import os
import sys

REF_ROOT = os.environ.get("WRSN_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "rl_env"))


def load_reference():
    """Return (WRSN class, NetworkIO class, MobileCharger class) of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    for p in (REF_ROOT, HERE):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, HERE)          # stand-ins shadow any real simpy/gym
    os.chdir(REF_ROOT)
    from rl_env.WRSN import WRSN
    from physical_env.network.NetworkIO import NetworkIO
    from physical_env.mc.MobileCharger import MobileCharger
    return WRSN, NetworkIO, MobileCharger
