"""Test-only stand-in: the reference imports ``DictConfig`` for isinstance checks in helpers outside the path."""


class DictConfig(dict):
    pass
