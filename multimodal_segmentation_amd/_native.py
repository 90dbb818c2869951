"""ctypes binding of libmmseg_hip.so (the gfx950 kernel library, C ABI in include/mmseg_hip.h).

The header is the single source of truth: prototypes are parsed from it, so every declared symbol must be
exported by the library (checked at load time).  There is NO fallback: if the library is missing or a symbol is
absent the import of any compute path raises.  Tensors cross the boundary as raw device pointers + sizes.
"""
import ctypes
import os
import re
import subprocess

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_PKG_DIR, 'csrc')
# MMSEG_HIP_LIB: another build of the same library (A/B measurements of kernel variants); the default is the in-tree build
LIB_PATH = os.environ.get('MMSEG_HIP_LIB') or os.path.join(CSRC_DIR, 'libmmseg_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), 'include', 'mmseg_hip.h')
SOURCES = ('conv.hip', 'pointwise.hip', 'norm.hip', 'act16.hip', 'dense.hip', 'tps.hip', 'augment.hip', 'loss.hip', 'pairloss.hip', 'optim.hip')

_CTYPES = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'void*': ctypes.c_void_p,
           'const float*': ctypes.c_void_p, 'float*': ctypes.c_void_p, 'const int*': ctypes.c_void_p, 'const void*': ctypes.c_void_p,
           'const long long*': ctypes.c_void_p}


class NativeLibraryError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtype names])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int|long)\s+(mmseg_\w+)\s*\(([^)]*)\)\s*;', text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                t = re.sub(r'\s*\w+$', '', a).strip() if not a.endswith('*') else a   # drop the parameter name
                t = re.sub(r'\s+', ' ', t).replace(' *', '*')
                types.append(t)
        protos[name] = (ret, types)
    return protos


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one in-tree shared library (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC_DIR, h) for h in ('common.hpp', 'smallconv.hpp', 'conv16.hpp', 'wgrad32h.hpp', 's2conv.hpp')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-o', LIB_PATH] + srcs
    if os.environ.get('MMSEG_AB_BUILD') == '1':      # measurement build: the kernel-selection switches of conv.hip read the environment
        cmd.insert(1, '-DMMSEG_AB')
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None
_protos = None


def load():
    """Load the library and type every symbol the header declares.  Raises NativeLibraryError loudly."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError('libmmseg_hip.so not found at %s -- run __graft_entry__.build() '
                                 '(there is no CPU/PyTorch fallback for the compute path)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    missing = [n for n in protos if not hasattr(lib, n)]
    if missing:
        raise NativeLibraryError('libmmseg_hip.so does not export: %s' % ', '.join(missing))
    for name, (ret, types) in protos.items():
        fn = getattr(lib, name)
        fn.restype = _CTYPES[ret]
        fn.argtypes = [_CTYPES[t] for t in types]
    _lib, _protos = lib, protos
    return lib


def _stream_handle(device):
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t, device, ctype='float*'):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('expected a tensor or None, got %r' % type(t))
    if 'void' in ctype:        # a tensor operand of a `_t` entry point: fp32 or the 16-bit storage type, flagged by the caller
        if t.dtype not in (torch.float32, torch.bfloat16, torch.float16) or not t.is_contiguous() or t.device != device:
            raise ValueError('kernel operand (%s) must be a contiguous fp32 / bf16 / fp16 tensor on %s (got %s %s)' % (ctype, device, t.dtype, t.device))
        return t.data_ptr()
    want = torch.int64 if 'long long' in ctype else (torch.int32 if 'int' in ctype else torch.float32)
    if t.dtype != want or not t.is_contiguous() or t.device != device:
        raise ValueError('kernel operand (%s) must be a contiguous %s tensor on %s (got %s %s contiguous=%s)'
                         % (ctype, want, device, t.dtype, t.device, t.is_contiguous()))
    return t.data_ptr()


def call(name, *args):
    """Launch `name` on the current stream of the operands' device.  Tensor arguments become device pointers;
    the trailing `stream` parameter is appended automatically.  Raises on a non-zero hipError_t."""
    lib = load()
    ret, types = _protos[name]
    device = None
    for a in args:
        if isinstance(a, torch.Tensor):
            device = a.device
            break
    takes_stream = bool(types) and types[-1] == 'void*'
    if takes_stream:
        if device is None or device.type != 'cuda':
            raise NativeLibraryError('%s needs device tensors (got %s); the HIP path has no CPU fallback' % (name, device))
    conv = []
    for i, a in enumerate(args):
        is_ptr = a is None or isinstance(a, torch.Tensor)
        conv.append(_ptr(a, device, types[i] if i < len(types) else 'float*') if is_ptr else a)
    if takes_stream:
        conv.append(_stream_handle(device))
    if len(conv) != len(types):
        raise TypeError('%s expects %d arguments, got %d' % (name, len(types), len(conv)))
    rc = getattr(lib, name)(*conv)
    if takes_stream and rc != 0:
        raise NativeLibraryError('%s failed with hipError_t %d' % (name, rc))
    return rc
