"""hqp_amd -- MI355X-native interior-point KKT linear-system path of omuses/hqp.

Only the hot path lives here: the HIP kernels + C ABI (``csrc/``,
``libhqpkkt.so``), the host-side mirror of the reference's ``Hqp_IpMatrix``
plugin interface (``ipmatrix``) and synthetic ``Hqp_Program`` generators
(``problems``).
"""
from .problems import Program  # noqa: F401


def __getattr__(name):
    if name in ("Hqp_IpMatrix", "Hqp_IpSpBKP", "Hqp_IpRedSpBKP", "Hqp_IpLQDOCP", "IpSpBKP", "IpRedSpBKP", "IpLQDOCP",
                "SingularError", "KktError", "selftest_mfma"):
        from . import ipmatrix
        return getattr(ipmatrix, name)
    raise AttributeError(name)
