"""pywfa_amd — MI355X-native batched wavefront aligner behind pywfa's interface.

Same names as ``pywfa/__init__.py`` of the reference, plus ``AlignmentResult``.
"""
from .align import (WavefrontAligner, AlignmentResult, clip_cigartuples, cigartuples_to_str,  # noqa: F401
                    elide_mismatches_from_cigar)

__all__ = ["WavefrontAligner", "AlignmentResult", "clip_cigartuples", "cigartuples_to_str",
           "elide_mismatches_from_cigar"]
__version__ = "0.1.0"
