from .Config import Config, ConfigGlobal  # noqa: F401
