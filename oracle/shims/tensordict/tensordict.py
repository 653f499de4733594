from . import TensorDict, TensorDictBase  # noqa: F401  (the reference imports `tensordict.tensordict.TensorDict`)
