"""avatarclip_amd -- MI355X-native (gfx950) implementation of the AvatarCLIP AppearanceGen hot path."""
__version__ = "0.1.0"
