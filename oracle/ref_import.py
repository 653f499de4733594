"""ORACLE (test infrastructure): import the reference's hot-path source files VERBATIM.

The reference (ai4co/rl4co v0.6.0, mounted read-only at ``/root/reference``) cannot be imported
as a package in this container: its ``__init__`` files pull in lightning / hydra / wandb, and
``tensordict`` / ``torchrl`` are not installed. Its rollout arithmetic, however,
is stock ``torch``. This loader

  * registers EMPTY package modules for ``rl4co`` and its sub-packages whose ``__path__`` points
    at the reference checkout — so ``import rl4co.utils.ops`` executes the reference's own
    ``utils/ops.py`` byte for byte, while the heavyweight ``__init__.py`` files are skipped;
  * puts ``oracle/shims`` (inert ``tensordict`` / ``torchrl`` / ``lightning`` stand-ins holding no
    arithmetic) on ``sys.path``;
  * stubs the two ``render`` modules (matplotlib) the env files import at top level.

Nothing is copied from the reference. The reference tree only exists in the build container, so
this module is used by ``oracle/gen_golden.py`` and by the ``reference``-marked CPU tests, which
skip when ``/root/reference`` is absent (e.g. on the GPU box).
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types
from pathlib import Path

REFERENCE_ROOT = Path(os.environ.get("RL4CO_REFERENCE_ROOT", "/root/reference"))
SHIMS = Path(__file__).resolve().parent / "shims"

# sub-packages registered as empty shells (their __init__.py is NOT executed)
_SHELL_PACKAGES = [
    "rl4co",
    "rl4co.utils",
    "rl4co.data",
    "rl4co.envs",
    "rl4co.envs.common",
    "rl4co.envs.routing",
    "rl4co.envs.routing.tsp",
    "rl4co.envs.routing.cvrp",
    "rl4co.envs.routing.op",
    "rl4co.envs.routing.pctsp",
    "rl4co.envs.routing.pdp",
    "rl4co.envs.routing.cvrptw",
    "rl4co.envs.routing.spctsp",
    "rl4co.models",
    "rl4co.models.nn",
    "rl4co.models.nn.graph",
    "rl4co.models.common",
    "rl4co.models.zoo",
    "rl4co.models.zoo.am",
    "rl4co.models.rl",
    "rl4co.models.rl.common",
    "rl4co.models.rl.reinforce",
    "rl4co.tasks",
    "rl4co.models.zoo.pomo",
]

# names the reference imports from a package's __init__ -> module that really defines them
_LAZY = {
    "rl4co.utils": {"get_pylogger": "rl4co.utils.pylogger"},
    "rl4co.envs": {
        "RL4COEnvBase": "rl4co.envs.common.base",
        "TSPEnv": "rl4co.envs.routing.tsp.env",
        "CVRPEnv": "rl4co.envs.routing.cvrp.env",
        "OPEnv": "rl4co.envs.routing.op.env",
        "PCTSPEnv": "rl4co.envs.routing.pctsp.env",
        "PDPEnv": "rl4co.envs.routing.pdp.env",
        "CVRPTWEnv": "rl4co.envs.routing.cvrptw.env",
        "SPCTSPEnv": "rl4co.envs.routing.spctsp.env",
    },
    "rl4co.models.zoo.am": {"AttentionModelPolicy": "rl4co.models.zoo.am.policy"},
}


class _Shell(types.ModuleType):
    """Empty package whose selected attributes resolve lazily to the reference's own modules."""

    def __getattr__(self, name):
        lazy = _LAZY.get(self.__name__, {})
        if name in lazy:
            value = getattr(importlib.import_module(lazy[name]), name)
            setattr(self, name, value)
            return value
        if self.__name__ == "rl4co.envs" and name == "get_env":
            return _get_env
        raise AttributeError(f"{self.__name__!r} shell has no attribute {name!r}")


def _get_env(env_name: str, *args, **kwargs):
    """rl4co/envs/__init__.py:65-84 restricted to the environments on the path."""
    envs = sys.modules["rl4co.envs"]
    registry = {"tsp": "TSPEnv", "cvrp": "CVRPEnv", "op": "OPEnv", "pctsp": "PCTSPEnv", "pdp": "PDPEnv", "cvrptw": "CVRPTWEnv", "spctsp": "SPCTSPEnv"}
    if env_name not in registry:
        raise ValueError(f"Unknown environment {env_name}. Available (oracle shell): {list(registry)}")
    return getattr(envs, registry[env_name])(*args, **kwargs)


def available() -> bool:
    return (REFERENCE_ROOT / "rl4co" / "utils" / "ops.py").is_file()


_installed = False


def install() -> None:
    """Make ``import rl4co.<hot path module>`` load the reference files verbatim. Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference checkout not found under {REFERENCE_ROOT}")
    for name in ("tensordict", "torchrl", "lightning", "omegaconf"):
        try:  # never shadow a real installation
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        if spec is not None and str(SHIMS) not in str(spec.origin):
            raise RuntimeError(f"{name} is installed for real; the shims are not needed — import rl4co directly")
    sys.path.insert(0, str(SHIMS))
    for pkg in _SHELL_PACKAGES:
        mod = _Shell(pkg)
        mod.__path__ = [str(REFERENCE_ROOT / pkg.replace(".", "/"))]
        mod.__package__ = pkg
        sys.modules[pkg] = mod
        if "." in pkg:
            parent, _, child = pkg.rpartition(".")
            setattr(sys.modules[parent], child, mod)
    for env in ("tsp", "cvrp", "op", "pctsp", "pdp", "cvrptw"):  # matplotlib renderers: not on the path, not installed
        stub = types.ModuleType(f"rl4co.envs.routing.{env}.render")
        stub.render = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("render is out of scope"))
        stub.render_improvement = stub.render
        sys.modules[stub.__name__] = stub
    _installed = True


def load():
    """Return a namespace with the reference's own classes/functions for the rollout path."""
    install()
    ns = types.SimpleNamespace()
    ns.ops = importlib.import_module("rl4co.utils.ops")
    ns.decoding = importlib.import_module("rl4co.utils.decoding")
    ns.TSPEnv = importlib.import_module("rl4co.envs.routing.tsp.env").TSPEnv
    ns.CVRPEnv = importlib.import_module("rl4co.envs.routing.cvrp.env").CVRPEnv
    ns.OPEnv = importlib.import_module("rl4co.envs.routing.op.env").OPEnv
    ns.PCTSPEnv = importlib.import_module("rl4co.envs.routing.pctsp.env").PCTSPEnv
    ns.PDPEnv = importlib.import_module("rl4co.envs.routing.pdp.env").PDPEnv
    ns.CVRPTWEnv = importlib.import_module("rl4co.envs.routing.cvrptw.env").CVRPTWEnv
    ns.SPCTSPEnv = importlib.import_module("rl4co.envs.routing.spctsp.env").SPCTSPEnv
    ns.TSPGenerator = importlib.import_module("rl4co.envs.routing.tsp.generator").TSPGenerator
    ns.CVRPGenerator = importlib.import_module("rl4co.envs.routing.cvrp.generator").CVRPGenerator
    ns.AttentionModelPolicy = importlib.import_module("rl4co.models.zoo.am.policy").AttentionModelPolicy
    ns.attention = importlib.import_module("rl4co.models.nn.attention")
    ns.TensorDict = importlib.import_module("tensordict").TensorDict
    return ns
