from .gumbel_sampler import GumbelSoftmaxSampler  # noqa: F401
from .uniform_sampler import UniformSampler  # noqa: F401
