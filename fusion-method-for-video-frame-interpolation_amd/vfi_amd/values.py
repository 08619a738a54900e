"""DecompValues: the (high_level, phase, amplitude, low_level) pyramid representation.

Field ORDER follows reference src/train/pyramid.py:12-18 (callers use it positionally)."""
from collections import namedtuple

DecompValues = namedtuple("values", "high_level, phase, amplitude, low_level")


class NormalizedValues(DecompValues):
    """What PhaseNet.normalize_vals returns: the same four fields (normalised), plus `concat`: per level
    the PhaseNet block's input buffer whose phase / amplitude channels are already filled (the `phase`
    and `amplitude` fields are views into it), so PhaseNet.forward needs no concat copies."""
    concat = None
    low_concat = None
