from ao_amd.pointops2 import pointops  # noqa: F401
