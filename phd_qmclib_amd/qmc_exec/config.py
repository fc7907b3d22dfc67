"""Configuration files of the command-line front end (reference:
qmc_exec/config.py:23-108): YAML (or TOML) with a `meta` table and an
`app_spec` list of {proc, proc_input, proc_output, proc_id}; relative file
locations are relative to the configuration file."""
import pathlib
import typing as t
from collections.abc import Sequence

__all__ = ['Loader', 'CONFIG_FILE_EXTENSIONS']

CONFIG_FILE_EXTENSIONS = ('.yml', '.yaml', '.toml')
YAML_EXTENSIONS = ('.yml', '.yaml')


class Loader:
    """Load the configuration for a set of QMC procedures."""

    def __init__(self, io_file_handler_types: t.Tuple[str, ...] = ('HDF5_FILE',),
                 file_extensions: t.Tuple[str, ...] = CONFIG_FILE_EXTENSIONS):
        self.io_file_handler_types = tuple(io_file_handler_types)
        self.file_extensions = tuple(file_extensions)

    def load(self, location: t.Union[str, pathlib.Path]):
        path = pathlib.Path(location)
        suffix = path.suffix
        if not suffix:
            raise IOError('config file has no extension')
        if suffix not in self.file_extensions:
            raise IOError('unknown file extension')
        if suffix in YAML_EXTENSIONS:
            import yaml
            with path.open('r', encoding='utf-8') as fp:
                config_data = yaml.safe_load(fp)
        else:
            try:
                import tomllib as toml_reader          # Python >= 3.11
            except ImportError:
                import tomli as toml_reader
            with path.open('rb') as fp:
                config_data = toml_reader.load(fp)
        # Keep support for old config files.
        if 'main_proc_set' in config_data:
            config_data['app_spec'] = config_data.pop('main_proc_set')
        app_spec_data = config_data['app_spec']
        if isinstance(app_spec_data, Sequence) and \
                not isinstance(app_spec_data, (str, bytes)):
            app_spec_config_set = [dict(c) for c in app_spec_data]
        else:
            app_spec_config_set = [dict(app_spec_data)]
        loc_parent = path.absolute().parent
        for app_spec_conf in app_spec_config_set:
            self.fix_app_spec_locations(app_spec_conf, loc_parent)
        config_data['app_spec'] = app_spec_config_set
        return config_data

    def fix_app_spec_locations(self, app_spec_config: t.MutableMapping,
                               config_path: pathlib.Path):
        """Relative paths are relative to the configuration file."""
        # deprecated aliases
        if 'input' in app_spec_config:
            app_spec_config['proc_input'] = app_spec_config.pop('input')
        if 'output' in app_spec_config:
            app_spec_config['proc_output'] = app_spec_config.pop('output')
        for key in ('proc_input', 'proc_output'):
            handler = dict(app_spec_config[key])
            if handler['type'] in self.io_file_handler_types:
                # an absolute location discards config_path by itself
                handler['location'] = str(config_path / handler['location'])
            app_spec_config[key] = handler
