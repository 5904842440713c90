"""Task-side consumers of the front-end kernels (aps/task): spectral approximation objectives and
the maximum-likelihood enhancement objective."""
from aps_amd.task.base import Task  # noqa: F401
