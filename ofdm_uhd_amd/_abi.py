"""ctypes view of include/ofdm_hip.h and the loader for libofdm_hip.so.

The library is the product: there is no Python/NumPy fallback.  Loading fails
loudly if the shared object has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C ofdm_uhd_amd/csrc``).
"""
import ctypes as C
import os

OFDM_ABI_VERSION = 5
OFDM_MAX_FFT = 4096
OFDM_MAX_TAPS = 512
OFDM_MAX_ARITY = 256
OFDM_MASK_LEN = 4096
OFDM_MAX_PKT_LEN = 4096
OFDM_MAX_CARRIER_HEX = 1024

OFDM_OK = 0
OFDM_E_INVAL = -1
OFDM_E_NOMEM = -2
OFDM_E_CAPACITY = -3
OFDM_E_HIP = -4
OFDM_E_OVERFLOW = -5

OFDM_F_DEVICE_PTRS = 1 << 0
OFDM_F_PAD_FOR_USRP = 1 << 1

(TAP_TX_PACKETS, TAP_TX_FREQ, TAP_RX_CHAN_FILT, TAP_RX_METRIC, TAP_RX_PEAKS, TAP_RX_ANGLES,
 TAP_RX_FRAMES, TAP_RX_FFT, TAP_RX_ACQ, TAP_RX_SINK, TAP_RX_PACKETS, TAP_TX_MAPPER, TAP_TX_IFFT, TAP_RX_SAMPLER,
 TAP_RX_SIGMIX, TAP_RX_NCO, TAP_RX_PRESEL, TAP_RX_DEMAPPED, TAP_COUNT) = range(19)
SYNC_PN, SYNC_FIXED = 0, 1

(K_FRAME, K_TX, K_CHAN, K_SYNC, K_PEAK, K_DEMOD, K_DEFRAME, K_SENSE, K_FILTER, K_EXACT, K_FRONT, K_COUNT) = range(12)
OFDM_SENSE_MAX_FFT = 4096


class ofdm_c32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class ofdm_cfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device_id", C.c_int32),
        ("flags", C.c_uint32),
        ("fft_length", C.c_uint32),
        ("occupied_tones", C.c_uint32),
        ("cp_length", C.c_uint32),
        ("arity", C.c_uint32),
        ("constellation", ofdm_c32 * OFDM_MAX_ARITY),
        ("known_symbol", ofdm_c32 * OFDM_MAX_FFT),
        ("tx_amplitude", C.c_float),
        ("phase_gain", C.c_float),
        ("freq_gain", C.c_float),
        ("eq_gain", C.c_float),
        ("max_fft_shift_len", C.c_uint32),
        ("sampler_timeout", C.c_uint32),
        ("peak_rise", C.c_float),
        ("peak_fall", C.c_float),
        ("peak_alpha", C.c_float),
        ("ntaps", C.c_uint32),
        ("taps", C.c_float * OFDM_MAX_TAPS),
        ("whitening_mask", C.c_uint8 * OFDM_MASK_LEN),
        ("whitener_offset", C.c_uint32),
        ("pad_seed", C.c_uint64),
        ("carrier_map", C.c_char * (OFDM_MAX_CARRIER_HEX + 8)),
        ("sync_mode", C.c_uint32),
        ("fixed_nsymbols", C.c_uint32),
        ("fixed_freq_offset", C.c_float),
        ("reserved0", C.c_uint32),
    ]


class ofdm_chan(C.Structure):
    _fields_ = [
        ("sigma", C.c_float),
        ("cfo", C.c_float),
        ("seed", C.c_uint64),
        ("stream_id", C.c_uint64),
        ("lead_samples", C.c_uint64),
        ("tail_samples", C.c_uint64),
    ]


class ofdm_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "symbols", "samples", "peaks", "frames", "headers_ok", "packets", "crc_ok",
        "chained_frames", "overflow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class ofdm_sense_cfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("fft_size", C.c_uint32),
        ("tune_delay", C.c_uint32),
        ("dwell_delay", C.c_uint32),
        ("avg_msgs", C.c_uint32),
        ("skip_msgs", C.c_uint32),
        ("threshold", C.c_double),
        ("window", C.c_float * OFDM_SENSE_MAX_FFT),
    ]


# every symbol include/ofdm_hip.h declares (tests check the .so exports all of them)
EXPORTS = (
    "ofdm_abi_version", "ofdm_device_count", "ofdm_create", "ofdm_destroy", "ofdm_last_error",
    "ofdm_set_stream", "ofdm_set_tx_amplitude", "ofdm_set_carrier_map", "ofdm_set_channel", "ofdm_framed_len",
    "ofdm_make_packets", "ofdm_tx_frame_count", "ofdm_tx", "ofdm_tx_async", "ofdm_wait", "ofdm_channel", "ofdm_rx",
    "ofdm_set_taps", "ofdm_tap", "ofdm_prof_enable", "ofdm_prof_reset", "ofdm_prof_get",
    "ofdm_kernel_name", "ofdm_sense_count", "ofdm_sense", "ofdm_sense_decide", "ofdm_set_rx_sense",
    "ofdm_rx_sense_result", "ofdm_sense_device_msgs", "ofdm_sense_redecide",
    "ofdm_rx_packet_pos", "ofdm_rx_nco_state", "ofdm_rx_set_flag_history", "ofdm_rx_set_origin", "ofdm_rx_submit", "ofdm_rx_snr",
)

_LIB = None
# OFDM_HIP_LIB selects another build of the same library (e.g. the -DSYNC_STAMPS diagnostic build)
LIB_PATH = os.environ.get("OFDM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                         "libofdm_hip.so")


def _declare(lib):
    vp, u8p, u32p, u64p = C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    H = C.c_void_p
    lib.ofdm_abi_version.restype = C.c_int
    lib.ofdm_device_count.restype = C.c_int
    lib.ofdm_create.argtypes = [C.POINTER(ofdm_cfg), C.POINTER(H)]
    lib.ofdm_destroy.argtypes = [H]
    lib.ofdm_destroy.restype = None
    lib.ofdm_last_error.argtypes = [H]
    lib.ofdm_last_error.restype = C.c_char_p
    lib.ofdm_set_stream.argtypes = [H, vp]
    lib.ofdm_set_tx_amplitude.argtypes = [H, C.c_float]
    lib.ofdm_set_carrier_map.argtypes = [H, C.c_char_p]
    lib.ofdm_set_channel.argtypes = [H, C.POINTER(ofdm_chan)]
    lib.ofdm_framed_len.argtypes = [H, C.c_uint32, u32p]
    lib.ofdm_make_packets.argtypes = [H, u8p, u64p, u32p, C.c_int, u8p, C.c_uint64, u64p]
    lib.ofdm_tx_frame_count.argtypes = [H, u32p, C.c_int, u64p, u64p]
    lib.ofdm_tx.argtypes = [H, u8p, u64p, u32p, C.c_int, vp, C.c_uint64, u64p, C.POINTER(ofdm_stats)]
    lib.ofdm_tx_async.argtypes = lib.ofdm_tx.argtypes
    lib.ofdm_wait.argtypes = [H]
    lib.ofdm_channel.argtypes = [H, vp, C.c_uint64, C.POINTER(ofdm_chan), C.c_uint64]
    lib.ofdm_rx.argtypes = [H, vp, C.c_uint64, u8p, C.c_uint64, u64p, u32p, C.POINTER(C.c_uint8),
                            C.c_int, C.POINTER(C.c_int), C.POINTER(ofdm_stats)]
    lib.ofdm_set_taps.argtypes = [H, C.c_uint32]
    lib.ofdm_tap.argtypes = [H, C.c_int, vp, C.c_uint64, u64p]
    lib.ofdm_prof_enable.argtypes = [H, C.c_int]
    lib.ofdm_prof_reset.argtypes = [H]
    lib.ofdm_prof_get.argtypes = [H, C.c_int, C.POINTER(C.c_double), u64p]
    lib.ofdm_kernel_name.argtypes = [C.c_int]
    lib.ofdm_kernel_name.restype = C.c_char_p
    SC = C.POINTER(ofdm_sense_cfg)
    lib.ofdm_sense_count.argtypes = [SC, C.c_uint64, u64p, u64p]
    lib.ofdm_sense.argtypes = [H, SC, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p, u64p]
    lib.ofdm_sense_decide.argtypes = [H, SC, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p]
    lib.ofdm_set_rx_sense.argtypes = [H, SC]
    lib.ofdm_rx_sense_result.argtypes = [H, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p, u64p]
    lib.ofdm_rx_packet_pos.argtypes = [H, vp, C.c_int, C.POINTER(C.c_int)]
    lib.ofdm_rx_nco_state.argtypes = [H, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.ofdm_rx_set_origin.argtypes = [H, C.c_uint64]
    lib.ofdm_rx_snr.argtypes = [H, C.POINTER(C.c_float)]
    lib.ofdm_rx_submit.argtypes = [H, C.c_void_p, C.c_uint64]
    lib.ofdm_rx_set_flag_history.argtypes = [H, C.c_int, C.c_int, vp, vp, vp, C.c_int64, C.c_int64, C.c_uint64, C.c_double]
    lib.ofdm_sense_device_msgs.argtypes = [H, C.POINTER(C.c_void_p), u64p, u32p]
    lib.ofdm_sense_redecide.argtypes = [H, SC]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("ofdm_abi_version", "ofdm_device_count"):
            pass
    return lib


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the name libofdm_hip.so needs).  If that copy is mapped first the dynamic linker
    hands it to us as well; if /opt/rocm's copy came first, a later ``import torch`` would map a SECOND
    runtime and find "No HIP GPUs".  So when torch is installed, map its runtime before ours --
    without importing torch."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load libofdm_hip.so (once).  Raises ImportError if it is missing: the HIP
    engine IS the implementation, nothing falls back to the CPU."""
    global _LIB
    if _LIB is None:
        _preload_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libofdm_hip.so not found at %s -- build it first (__graft_entry__.build() or "
                "`make -C ofdm_uhd_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        ver = lib.ofdm_abi_version()
        if ver != OFDM_ABI_VERSION:
            raise ImportError("libofdm_hip.so ABI %d != expected %d" % (ver, OFDM_ABI_VERSION))
        _LIB = _declare(lib)
    return _LIB
