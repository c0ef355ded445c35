"""Drop-in name for the reference's ``pyviz/apap_utils.py`` module.

``from apap_utils import *`` in the reference pulls in four helpers; they live in
:mod:`cvx_proj_amd.geometry` here.
"""
from .geometry import final_size, get_mesh, get_vertice, uniform_blend

__all__ = ["get_mesh", "get_vertice", "final_size", "uniform_blend"]
