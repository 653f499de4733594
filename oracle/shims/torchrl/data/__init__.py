"""Stand-in for ``torchrl.data`` spec classes: inert records (descriptive only on this path)."""


class _Spec:
    def __init__(self, *args, **kwargs):
        self.args = args
        self.kwargs = kwargs


class Bounded(_Spec):
    pass


class Unbounded(_Spec):
    pass


class Composite(_Spec):
    pass


BoundedTensorSpec = Bounded
UnboundedContinuousTensorSpec = Unbounded
UnboundedDiscreteTensorSpec = Unbounded
CompositeSpec = Composite
