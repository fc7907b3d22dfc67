"""Application layer of the command-line front end (reference:
qmc_exec/cli_app.py:12-122, mrbp_qmc/{vmc,dmc}_exec/cli_app.py): an `AppSpec`
binds a `Proc` to where its input comes from and where its result goes; a
`CLIApp` runs a list of them in order."""
import typing as t

import attr

from . import exec_logger

__all__ = ['AppMeta', 'make_app_classes']


def proc_cli_tags_converter(tag_or_tags: t.Union[str, t.Sequence[str]]):
    if isinstance(tag_or_tags, str):
        return tag_or_tags
    return ' - '.join('#' + str(tag) for tag in tag_or_tags)


_str = attr.validators.instance_of(str)


@attr.s(auto_attribs=True)
class AppMeta:
    """Metadata of the application."""
    name: str = attr.ib(validator=_str)
    description: str = attr.ib(validator=_str)
    author: str = attr.ib(validator=_str)
    author_email: str = attr.ib(validator=_str)
    institution: str = attr.ib(validator=_str)
    category: str = attr.ib(validator=_str)
    tags: str = attr.ib(converter=proc_cli_tags_converter, validator=_str)


def make_app_classes(proc_cls, proc_input_cls, model_sys_conf_cls,
                     handler_cls, model_sys_conf_type: str):
    """(AppSpec, CLIApp, get_io_handler) for one sampling type."""

    def get_io_handler(config: t.Mapping):
        handler_config = dict(config)
        handler_type = handler_config['type']
        if handler_type == model_sys_conf_type:
            return model_sys_conf_cls(**handler_config)
        if handler_type == 'HDF5_FILE':
            return handler_cls.from_config(handler_config)
        raise TypeError(f"unknown handler type {handler_type}")

    @attr.s(auto_attribs=True)
    class AppSpec:
        #: Procedure spec.
        proc: t.Any = attr.ib(validator=attr.validators.instance_of(proc_cls))
        #: Input spec.
        proc_input: t.Any = attr.ib(validator=attr.validators.instance_of(
            (model_sys_conf_cls, handler_cls)))
        #: Output spec.
        proc_output: t.Any = attr.ib(
            validator=attr.validators.instance_of(handler_cls))
        #: Procedure id.
        proc_id: t.Optional[int] = attr.ib(
            default=None,
            validator=attr.validators.optional(
                attr.validators.instance_of(int)))

        @classmethod
        def from_config(cls, config: t.Mapping):
            self_config = dict(config)
            proc = proc_cls.from_config(self_config['proc'])
            proc_id = self_config.get('proc_id', 0)
            if 'input' in self_config:
                self_config['proc_input'] = self_config.pop('input')
            if 'output' in self_config:
                self_config['proc_output'] = self_config.pop('output')
            input_handler = get_io_handler(self_config['proc_input'])
            output_handler = get_io_handler(self_config['proc_output'])
            if not isinstance(output_handler, handler_cls):
                raise TypeError('only the HDF5_FILE is supported as '
                                'output handler')
            return cls(proc=proc, proc_input=input_handler,
                       proc_output=output_handler, proc_id=proc_id)

        def build_input(self):
            proc_input = self.proc_input
            if isinstance(proc_input, model_sys_conf_cls):
                return proc_input_cls.from_model_sys_conf_spec(proc_input,
                                                               self.proc)
            if isinstance(proc_input, handler_cls):
                return proc_input_cls.from_result(proc_input.load(),
                                                  self.proc)
            raise TypeError

        def exec(self, dump_output: bool = True):
            proc_result = self.proc.exec(self.build_input())
            if dump_output:
                self.proc_output.dump(proc_result)
            return proc_result

    @attr.s(auto_attribs=True)
    class CLIApp:
        """Entry point for the CLI."""
        meta: AppMeta
        app_spec: t.Sequence[AppSpec] = attr.ib(
            validator=attr.validators.instance_of((list, tuple)))

        @classmethod
        def from_config(cls, config: t.Mapping):
            self_config = dict(config.items())
            app_meta = AppMeta(**self_config['meta'])
            app_spec_set = []
            for proc_num, app_spec_config in \
                    enumerate(self_config.pop('app_spec')):
                app_spec_config = dict(app_spec_config)
                proc_id = app_spec_config.get('proc_id', None)
                app_spec_config['proc_id'] = \
                    proc_num if proc_id is None else proc_id
                app_spec_set.append(AppSpec.from_config(app_spec_config))
            return cls(meta=app_meta, app_spec=app_spec_set)

        def exec(self, dump_output: bool = True):
            results = []
            n = len(self.app_spec)
            exec_logger.info(f'Starting the execution of a set of {n} QMC '
                             f'calculations...')
            for proc_num, app_spec in enumerate(self.app_spec, 1):
                exec_logger.info(f'Starting procedure ID{proc_num}...')
                results.append(app_spec.exec(dump_output))
                exec_logger.info(f'Procedure ID{proc_num} completed.')
            exec_logger.info('All the QMC calculations have completed.')
            return results

    return AppSpec, CLIApp, get_io_handler
