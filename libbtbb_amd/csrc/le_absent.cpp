// le_absent.cpp -- the Bluetooth LE half of libbtbb's ABI (lib/src/btbb.h:229-281, `lell_*`) is NOT part
// of this library: it lies outside the BR/EDR baseband path this repository rebuilds (SURVEY.md section 2,
// rows marked out of scope).  The shared object nevertheless carries libbtbb's SONAME, so a program that
// was linked against the reference and uses LE would otherwise bind lazily and die somewhere inside a
// capture with "symbol lookup error".  Every LE entry point is therefore exported and fails LOUDLY and
// immediately, naming the remedy; nothing here pretends to work.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static void le_absent(const char *symbol)
{
	fprintf(stderr,
		"libbtbb (MI355X HIP build): %s() was called, but the Bluetooth LE half of libbtbb (lell_*) is not\n"
		"part of this library -- it provides the BR/EDR baseband path only.  Link the reference libbtbb's\n"
		"bluetooth_le_packet / LE pcap objects for LE captures.\n", symbol);
	abort();
}

#define LE_EXPORT __attribute__((visibility("default")))
#define LE_ABSENT(ret, name, args) extern "C" LE_EXPORT ret name args { le_absent(#name); return (ret)0; }
#define LE_ABSENT_VOID(name, args) extern "C" LE_EXPORT void name args { le_absent(#name); }

struct lell_packet;
struct lell_pcapng_handle;
struct lell_pcap_handle;

LE_ABSENT_VOID(lell_allocate_and_decode, (const uint8_t *, uint16_t, uint32_t, lell_packet **))
LE_ABSENT(lell_packet *, lell_packet_new, (void))
LE_ABSENT_VOID(lell_packet_ref, (lell_packet *))
LE_ABSENT_VOID(lell_packet_unref, (lell_packet *))
LE_ABSENT(uint32_t, lell_get_access_address, (const lell_packet *))
LE_ABSENT(unsigned, lell_get_access_address_offenses, (const lell_packet *))
LE_ABSENT(unsigned, lell_packet_is_data, (const lell_packet *))
LE_ABSENT(unsigned, lell_get_channel_index, (const lell_packet *))
LE_ABSENT(unsigned, lell_get_channel_k, (const lell_packet *))
LE_ABSENT(const char *, lell_get_adv_type_str, (const lell_packet *))
LE_ABSENT_VOID(lell_print, (const lell_packet *))
LE_ABSENT(int, lell_pcapng_create_file, (const char *, const char *, lell_pcapng_handle **))
LE_ABSENT(int, lell_pcapng_append_packet, (lell_pcapng_handle *, const uint64_t, const int8_t, const int8_t, const uint32_t,
					    const lell_packet *))
LE_ABSENT(int, lell_pcapng_record_connect_req, (lell_pcapng_handle *, const uint64_t, const uint8_t *))
LE_ABSENT(int, lell_pcapng_close, (lell_pcapng_handle *))
LE_ABSENT(int, lell_pcap_create_file, (const char *, lell_pcap_handle **))
LE_ABSENT(int, lell_pcap_ppi_create_file, (const char *, int, lell_pcap_handle **))
LE_ABSENT(int, lell_pcap_append_packet, (lell_pcap_handle *, const uint64_t, const int8_t, const int8_t, const uint32_t,
					  const lell_packet *))
LE_ABSENT(int, lell_pcap_append_ppi_packet, (lell_pcap_handle *, const uint64_t, const uint8_t, const int8_t, const int8_t,
					      const int8_t, const uint8_t, const lell_packet *))
LE_ABSENT(int, lell_pcap_close, (lell_pcap_handle *))
