"""Minimal stand-in for `gymnasium`, used ONLY by oracle/capture/capture.py.

gymnasium is not installable in the build container (no network).  The reference
env classes need nothing from it beyond a base class and two space
descriptors, so this stub provides exactly those names.  Test infrastructure:
never imported by the product package `beacon_amd`.
"""
from . import spaces  # noqa: F401


class Env(object):
    metadata = {}
