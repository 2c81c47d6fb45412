"""Synthetic inputs shared by the tests and the golden generator (re-export)."""
from ao_amd.synth import *  # noqa: F401,F403
from ao_amd.synth import lattice_cloud, random_cloud, room_cloud, room_scene, scene_batch  # noqa: F401
