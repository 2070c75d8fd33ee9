from taichi_glsl import normalize, dot, cross, length, reflect      # noqa: F401  (UtilsFunc.py: from taichi_glsl.vector import normalize)
