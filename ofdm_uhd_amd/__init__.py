"""ofdm_uhd_amd -- MI355X-native OFDM TX/RX engine behind the ofdm_mod /
ofdm_demod API of rubiruchi/ofdm_uhd.  The DSP lives in csrc/ (hand-written HIP
for gfx950, C ABI in include/ofdm_hip.h); this package is the Python host side
that mirrors the reference's modules."""
__all__ = ["ofdm", "ofdm_packet_utils", "psk", "qam", "transmit_path", "receive_path",
           "engine", "config", "options"]
