def compile_mode(mode):
    """e3nn.util.jit.compile_mode only tags a class for e3nn's optional TorchScript compiler, which the reference never
    invokes."""
    def deco(cls):
        cls._e3nn_compile_mode = mode
        return cls
    return deco
