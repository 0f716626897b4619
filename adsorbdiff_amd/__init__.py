"""MI355X-native denoising-diffusion sampling path of AdsorbDiff (PaiNN score model +
reverse-SDE stepper) behind a C ABI.  See DESIGN.md / INTEGRATION.md."""

__version__ = "0.5.0"
