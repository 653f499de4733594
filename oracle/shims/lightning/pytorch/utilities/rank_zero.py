"""Stand-in for lightning's ``rank_zero_only``: single-process identity decorator."""


def rank_zero_only(fn):
    return fn
