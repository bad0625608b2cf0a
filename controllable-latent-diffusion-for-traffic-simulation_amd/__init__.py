"""MI355X-native latent-diffusion sampling path for CLD traffic simulation.

Host-side mirror of the reference call surface (`DmModel.forward / x_Tminus1 /
log_prob`, `LSTMVAE.lstm_dec`, `VaeModel.convert_action_to_state_and_action`)
over a C-ABI HIP library (`include/cld.h`, `csrc/`).  Import as `cld_amd`.

Importing the package never touches the GPU; constructing an `Engine` /
`DmModel` does, and fails loudly when libcld_hip.so or the device is missing.
"""
__version__ = "0.1.0"
