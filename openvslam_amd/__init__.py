"""openvslam_amd -- MI355X-native implementation of OpenVSLAM's per-frame hot path (ORB extraction, 256-bit Hamming
matching, local-BA linearisation) behind the C ABI of include/ovslam_hip.h. See DESIGN.md / INTEGRATION.md."""
