"""Loader of the HIP C-ABI library (include/hrl_envs.h).  There is no CPU fallback: a missing library or a
missing GPU is an error."""
import ctypes as C
import os

from . import _capi as K

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HRL_ENVS_LIB') or os.path.join(_PKG, 'libhrl_envs_hip.so')   # HRL_ENVS_LIB: another build of the same ABI (A/B measurements)
_lib = None

# every symbol include/hrl_envs.h declares
SYMBOLS = ['hrl_default_config', 'hrl_obs_dim', 'hrl_act_dim', 'hrl_items_stride', 'hrl_create', 'hrl_destroy', 'hrl_reset', 'hrl_step',
           'hrl_get_state', 'hrl_set_state', 'hrl_set_goals', 'hrl_next_target', 'hrl_last_error', 'hrl_backend', 'hrl_buffers_init', 'hrl_observe', 'hrl_update_config']


class HrlError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HrlError(f'{LIB_PATH} is missing: build it with `python -m hrl_pybullet_envs_amd.build` '
                           '(hipcc --offload-arch=gfx950); there is no CPU fallback')
        L = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(L, s)  # AttributeError if the library does not export the ABI
        L.hrl_last_error.restype = C.c_char_p
        L.hrl_backend.restype = C.c_char_p
        L.hrl_create.argtypes = [C.POINTER(K.hrl_config), C.POINTER(C.c_void_p)]
        L.hrl_destroy.argtypes = [C.c_void_p]
        L.hrl_update_config.argtypes = [C.c_void_p, C.POINTER(K.hrl_config), C.c_void_p]
        L.hrl_reset.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_void_p]
        L.hrl_step.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p]
        L.hrl_observe.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_void_p]
        L.hrl_buffers_init.argtypes = [C.POINTER(K.hrl_buffers)]
        L.hrl_get_state.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_void_p, C.c_void_p]
        L.hrl_set_state.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_void_p, C.c_void_p]
        L.hrl_set_goals.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.hrl_next_target.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_void_p, C.c_void_p]
        L.hrl_default_config.argtypes = [C.c_int32, C.POINTER(K.hrl_config)]
        L.hrl_obs_dim.argtypes = [C.POINTER(K.hrl_config)]
        L.hrl_act_dim.argtypes = [C.POINTER(K.hrl_config)]
        L.hrl_items_stride.argtypes = [C.POINTER(K.hrl_config)]
        _lib = L
    return _lib


def check(rc):
    if rc != K.HRL_OK:
        raise HrlError(f'hrl error {rc}: {lib().hrl_last_error().decode()}')


def default_config(kind, **over):
    cfg = K.hrl_config()
    check(lib().hrl_default_config(kind, C.byref(cfg)))
    names = {f[0] for f in K.hrl_config._fields_}
    mnames = {f[0] for f in K.hrl_model._fields_}
    for k, v in over.items():
        if (k[6:] not in mnames) if k.startswith('model_') else (k not in names):
            raise TypeError(f'hrl_config has no field {k!r}')  # a ctypes Structure would silently grow an attribute
        if k.startswith('model_'):
            setattr(cfg.model, k[6:], v)
        elif k in ('world_size', 'start_pos', 'walk_target', 'centroid_static_sum'):
            arr = getattr(cfg, k)
            for i, x in enumerate(v):
                arr[i] = x
        elif k == 'targets':
            if len(v) > K.HRL_MAX_TARGETS:
                raise ValueError(f'at most {K.HRL_MAX_TARGETS} targets')
            cfg.n_targets = len(v)
            for i, t in enumerate(v):
                cfg.targets[i][0], cfg.targets[i][1] = float(t[0]), float(t[1])
        else:
            setattr(cfg, k, v)
    return cfg
